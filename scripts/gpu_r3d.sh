#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8
python bench.py --config 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['kernel'], d['parity_check'], d['solve_only_M_not_emitted'])"
for op in solve solve_nom mm; do OP=$op TAG=spec python scripts/exp_config3.py 2>&1 | grep -v amdgpu.ids; done
OP=solve TAG=dense_chol RBD_EXP_NO_SPEC_CHOL=1 python scripts/exp_config3.py 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r3d.log 2>&1
cat gpurun_out/r3d.log
