#!/bin/bash
# walk kernels with the plan records back in LDS: parity, inverse dynamics and dynamics timings
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu -k "inverse or bias or walk or rnea or pair" 2>&1 | tail -3
for cfg in "f64 65536" "f32 65536"; do set -- $cfg; echo "$cfg $(timeout 600 python scripts/bench_ops.py --dtype $1 --batch $2 --only dynamics 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c60-400)"; done
