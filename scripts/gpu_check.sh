#!/bin/bash
# One gpurun call: GPU parity tests, smoke, bench, rocprof kernel trace. Outputs under gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== rocminfo" > gpurun_out/env.log; (rocm-smi --showproductname 2>&1 | head -20) >> gpurun_out/env.log
echo "== pytest gpu"; timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 | tee gpurun_out/pytest_gpu.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/smoke.log
echo "== bench"; timeout 600 python bench.py 2>&1 | tail -3 | tee gpurun_out/bench.log
echo "== bench variants"
for extra in "--graph" "--dtype f32" "--dtype f32 --batch 65536 --steps 200" "--batch 65536 --steps 200" "--layout soa" "--wrenches"; do
  echo "-- $extra"; timeout 600 python bench.py --no-cpu-baseline $extra 2>&1 | tail -1 | tee -a gpurun_out/bench_variants.log
done
echo "== rocprof"
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 500 > $GRAFT_REPO_ROOT/gpurun_out/rocprof_run.log 2>&1
cd $GRAFT_REPO_ROOT; find gpurun_out/prof -name "*stats*" | head; for f in $(find gpurun_out/prof -name "*kernel_stats.csv"); do head -8 $f; done
