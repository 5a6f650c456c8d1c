#!/bin/bash
# the compiled walk kernels: stress on other mechanisms, their tests, the walk tests around them
mkdir -p gpurun_out
{
timeout 800 python scripts/stress_walk_compiled.py 10 2>&1 | tail -9
timeout 900 python -m pytest tests/test_state_kernels.py -x -q -k "compiled_walk" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "walk or per_body or chains_scope or banked_f32_and_full" 2>&1 | tail -5
} > gpurun_out/r3f.log 2>&1
tail -40 gpurun_out/r3f.log
