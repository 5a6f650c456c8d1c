#!/bin/bash
# round-2 call G: walk kernels with scalar-loaded records: parity (walk tests), then the sweep
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest gpu (walk)"; timeout 1500 python -m pytest tests -x -q -m gpu -k "walk or pair or simulate" 2>&1 | tail -4
for dt in f64 f32; do
  timeout 300 python scripts/mapping_sweep.py --algos aba_walk --batches 8192,16384,65536 --dtypes $dt 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/walk_sweep_g.txt
done
