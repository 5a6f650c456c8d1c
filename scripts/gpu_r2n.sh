#!/bin/bash
# crba_kernel with the support-chain walk through an LDS table of records: parity, then mass_matrix! timings
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|Error" | tail -3
for cfg in "f64 4096" "f32 4096" "f64 65536"; do set -- $cfg; echo "$cfg $(timeout 600 python scripts/bench_ops.py --dtype $1 --batch $2 --only mass_matrix 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c60-300)"; done
