#!/bin/bash
# round-2 call A: GPU parity suite, state-kernel sweep (fp32, both layouts at the config-3 size), config-3 measurement pass
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest gpu"; timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/pytest_gpu.log
echo "== state sweep f32 aos"; timeout 600 python scripts/state_sweep.py f32 aos 2>&1 | tee gpurun_out/state_sweep_f32_aos.txt
echo "== state sweep f64 aos"; timeout 600 python scripts/state_sweep.py f64 aos 2>&1 | tee gpurun_out/state_sweep_f64_aos.txt
echo "== measure c3"; timeout 900 bash scripts/gpu_measure.sh 3 2>&1 | tail -5
