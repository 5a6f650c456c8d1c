#!/bin/bash
mkdir -p gpurun_out; rm -f gpurun_out/ablation.log
# the phase-ablation hooks are compiled only into this variant (hipcc is on the GPU box too)
OUT=librbd_hip_prof.so bash rigidbodydynamics.jl_amd/csrc/build.sh -DRBD_PROFILE_PHASES > /dev/null
for ph in 1 2 3 4 5 0; do
  RBD_LIB=$PWD/rigidbodydynamics.jl_amd/csrc/librbd_hip_prof.so RBD_ABA_STOP_AFTER=$ph python scripts/ablate_once.py 2>&1 | tail -1 | tee -a gpurun_out/ablation.log
done
