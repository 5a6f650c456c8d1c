#!/bin/bash
mkdir -p gpurun_out; rm -f gpurun_out/ablation.log
for ph in 1 2 3 4 5 0; do
  RBD_LIB=$PWD/rigidbodydynamics.jl_amd/csrc/librbd_hip_prof.so RBD_ABA_STOP_AFTER=$ph python scripts/ablate_once.py 2>&1 | tail -1 | tee -a gpurun_out/ablation.log
done
