#!/bin/bash
# the compiled walk kernel: its tests, the walk tests around it, and the fp64 large-batch entry points
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_state_kernels.py -x -q -k "compiled_walk" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "walk or chains_scope or banked_f32_and_full" 2>&1 | tail -5
timeout 600 python scripts/bench_ops.py --batch 65536 --dtype f64 --reps 30 --only "dynamics! (ABA)" 2>&1 | tail -3
timeout 600 python scripts/bench_ops.py --batch 65536 --dtype f64 --reps 30 --only "simulate" 2>&1 | tail -3
timeout 600 python scripts/bench_ops.py --batch 16384 --dtype f64 --reps 30 --only "simulate" 2>&1 | tail -3
} > gpurun_out/r3f.log 2>&1
tail -40 gpurun_out/r3f.log
