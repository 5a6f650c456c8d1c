#!/bin/bash
# the walk kernel compiled for the mechanism against the interpreting one, in one call
mkdir -p gpurun_out
{
echo "== compiled (fence between steps)"; timeout 600 python scripts/exp_walk_spec.py
echo "== compiled (no fence)"; RBD_JIT_FLAGS='-DRBD_WALK_STEP_FENCE()=' TIMING_ONLY=1 timeout 600 python scripts/exp_walk_spec.py | grep "^B "
echo "== interpreting"; RBD_SPEC_WALK_MIN_BATCH=1000000000 timeout 600 python scripts/exp_walk_spec.py | grep "^B "
} > gpurun_out/r3e.log 2>&1
tail -30 gpurun_out/r3e.log
