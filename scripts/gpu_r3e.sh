#!/bin/bash
# the walk kernels compiled for the mechanism against the interpreting ones, in one call
mkdir -p gpurun_out
{
echo "== compiled"; timeout 600 python scripts/exp_walk_spec.py
echo "== interpreting"; RBD_SPEC_WALK_MIN_BATCH=1000000000 timeout 600 python scripts/exp_walk_spec.py | grep "^B "
} > gpurun_out/r3e.log 2>&1
tail -40 gpurun_out/r3e.log
