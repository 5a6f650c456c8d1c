#!/bin/bash
# round-2 call E: walk kernels on the re-rooted plan: parity, then the walk kernel with and without re-rooting at 16384/65536, fp64 and packed fp32
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest gpu"; timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee gpurun_out/pytest_gpu.log
for dt in f64 f32; do
  echo "== rerooted $dt"; timeout 300 python scripts/mapping_sweep.py --algos aba_walk --batches 8192,16384,65536 --dtypes $dt 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/walk_reroot.txt
  echo "== pelvis-rooted $dt"; RBD_WALK_NO_REROOT=1 timeout 300 python scripts/mapping_sweep.py --algos aba_walk --batches 8192,16384,65536 --dtypes $dt 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/walk_reroot.txt
done
echo "== stress"; timeout 600 python scripts/stress_mappings.py 2>&1 | tail -5
