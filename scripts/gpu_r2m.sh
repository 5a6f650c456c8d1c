#!/bin/bash
# fp64 tile Cholesky on the matrix cores: whole GPU suite, then the fp64 routes that use it, with and without (RBD_NO_MFMA64=1)
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|Error" | tail -3
for B in 4096 65536; do
  echo "mfma64 B=$B $(timeout 600 python scripts/bench_ops.py --dtype f64 --batch $B --only Cholesky 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c60-330)"
  echo "reg    B=$B $(RBD_NO_MFMA64=1 timeout 600 python scripts/bench_ops.py --dtype f64 --batch $B --only Cholesky 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c60-330)"
done
