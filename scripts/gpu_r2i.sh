#!/bin/bash
# simulate with the quaternion-joint and element-wise stage work in one launch: parity, then µs per RK4 step
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu -k "simulate or contact or mk_stage or integr" 2>&1 | grep -E "passed|failed"
for dt in f64 f32; do for B in 4096 65536; do
  echo "$(python scripts/sim_prof.py $B $dt 2>&1 | grep 'us per step')"
done; done
