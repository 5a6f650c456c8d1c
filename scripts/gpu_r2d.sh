#!/bin/bash
# round-2 call D: RNEA walk kernel + simulate through the walk kernel: parity, then per-op timings at the BASELINE sizes
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest gpu"; timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee gpurun_out/pytest_gpu.log
for cfg in "f64 4096" "f64 65536" "f32 65536"; do
  set -- $cfg
  echo "== ops $1 B=$2"; timeout 600 python scripts/bench_ops.py --dtype $1 --batch $2 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/bench_ops_r02.jsonl
done
