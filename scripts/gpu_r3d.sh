#!/bin/bash
# round 3, second half: the kernels compiled per mechanism at run time against the kernels that interpret it — one gpurun call.
#   gpurun --timeout 1500 -- 'bash scripts/gpu_r3d.sh'   ->  gpurun_out/r3d.log
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
echo "== parity of both forms (tests/test_state_kernels.py, four-bar)"
timeout 1200 python -m pytest tests/test_state_kernels.py tests/test_gpu_parity.py -x -q -m gpu -k "state_kernels or compiled or four_bar" 2>&1 | tail -3
echo "== configs[2] and mass_matrix! alone, 65 536 fp32 states: compiled / interpreting (graph-replayed us per call)"
for op in solve solve_nom mm; do OP=$op TAG=compiled python scripts/exp_config3.py 2>&1 | grep -v amdgpu.ids; done
for op in solve mm; do OP=$op TAG=interpreting RBD_JIT=0 python scripts/exp_config3.py 2>&1 | grep -v amdgpu.ids; done
echo "== dynamics!, inverse_dynamics!: compiled / walk kernel"
TIMING_ONLY=1 python scripts/exp_aba_spec.py 2>&1 | grep -v amdgpu.ids
TIMING_ONLY=1 python scripts/exp_rnea_spec.py 2>&1 | grep -v amdgpu.ids | grep float32
echo "== four-bar (configs[4]): compiled / generic"
for j in 1 0; do RBD_JIT=$j python bench.py --config 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['roofline']['kernel'], round(d['ms_per_step']*1e3, 2), 'us per step', d['value'], 'evals/s')"; done
} > gpurun_out/r3d.log 2>&1
cat gpurun_out/r3d.log
