#!/bin/bash
# the role-pipelined mapping: parity tests, timing against the banked kernel, phase clocks
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests -x -q -m gpu -k "pipe" 2>&1 | grep -E "passed|failed"
for dt in f64 f32; do timeout 300 python scripts/mapping_sweep.py --algos aba_pipe,aba_banks --batches 16,1000,4096,8192 --dtypes $dt 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/pipe_sweep.txt; done
timeout 300 python scripts/mapping_sweep.py --algos aba_pipe,aba_banks --batches 4096 --dtypes f64 --wrenches 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/pipe_sweep.txt
export RBD_LIB=$PWD/rigidbodydynamics.jl_amd/csrc/librbd_hip_prof.so; python scripts/pipe_phases.py 2>&1 | grep -v amdgpu | tee gpurun_out/pipe_phases.txt
