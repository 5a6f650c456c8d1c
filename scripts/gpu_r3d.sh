#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
TIMING_ONLY=1 python scripts/exp_aba_spec.py 2>&1 | grep -v amdgpu.ids | tail -4
TIMING_ONLY=1 python scripts/exp_rnea_spec.py 2>&1 | grep -v amdgpu.ids | grep float32
timeout 900 python -m pytest tests/test_state_kernels.py -x -q -m gpu -k "compiled" 2>&1 | tail -3
} > gpurun_out/r3d.log 2>&1
cat gpurun_out/r3d.log
