#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
timeout 2400 python -m pytest tests -x -q -m gpu --durations=6 2>&1 | tail -16
python scripts/bench_ops.py --batch 65536 --dtype f32 2>&1 | grep -v amdgpu.ids | tail -30
} > gpurun_out/r3d.log 2>&1
cat gpurun_out/r3d.log
