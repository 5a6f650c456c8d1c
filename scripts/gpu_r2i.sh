#!/bin/bash
# simulate with the quaternion-joint stage launch on a side stream: parity, then µs per RK4 step with and without
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu -k "simulate or contact or mk_stage or integr" 2>&1 | tail -3
for dt in f64 f32; do for B in 4096 65536; do
  echo "side   $(python scripts/sim_prof.py $B $dt 2>&1 | grep 'us per step')"
  echo "serial $(RBD_NO_SIDE_STREAM=1 python scripts/sim_prof.py $B $dt 2>&1 | grep 'us per step')"
done; done
