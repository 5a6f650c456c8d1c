#!/bin/bash
# round-2 call C: whole GPU suite, smoke, then the measurement pass for every BASELINE config (bench + rocprofv3 stats + PMC) on the current sources
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest gpu"; timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee gpurun_out/pytest_gpu.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/smoke.log
echo "== measure"; timeout 1500 bash scripts/gpu_measure.sh 2 3 4 5 2>&1 | tail -12
echo "== sweep (defaults)"; timeout 600 python scripts/mapping_sweep.py --algos aba,aba_walk,aba_banks --batches 4096,8192,12288,16384,65536 2>&1 | grep -v amdgpu.ids | tee gpurun_out/walk_sweep.txt
for cfg in "f64 4096" "f64 65536" "f32 65536"; do
  set -- $cfg
  echo "== ops $1 B=$2"; timeout 600 python scripts/bench_ops.py --dtype $1 --batch $2 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/bench_ops_r02.jsonl
done
