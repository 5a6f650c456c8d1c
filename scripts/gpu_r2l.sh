#!/bin/bash
# banked kernel with the staged (single round trip) prologue: whole GPU suite, then the headline sweep
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|Error" | tail -3
for dt in f64 f32; do timeout 300 python scripts/mapping_sweep.py --algos aba_banks,aba_lanes --batches 64,1024,4096,8192 --dtypes $dt 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/bank_staged.txt; done
timeout 300 python scripts/mapping_sweep.py --algos aba_banks --batches 4096 --dtypes f64 --wrenches 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/bank_staged.txt
