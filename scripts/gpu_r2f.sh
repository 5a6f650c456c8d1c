#!/bin/bash
# round-2 call F: phase clocks of the walk kernel per track, re-rooted plan vs the original tree (profiling build made beforehand)
mkdir -p gpurun_out; export TMPDIR=/tmp
export RBD_LIB=$PWD/rigidbodydynamics.jl_amd/csrc/librbd_hip_prof.so
echo "== re-rooted"; timeout 300 python scripts/walk_phases.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/walk_phases_rr.txt
echo "== original tree"; RBD_WALK_NO_REROOT=1 timeout 300 python scripts/walk_phases.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/walk_phases_norr.txt
