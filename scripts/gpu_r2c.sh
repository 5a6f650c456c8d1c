#!/bin/bash
# round-2 call C: whole GPU suite, smoke, then the measurement pass for every BASELINE config (bench + rocprofv3 stats + PMC) on the current sources
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest gpu"; timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee gpurun_out/pytest_gpu.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/smoke.log
echo "== measure"; timeout 1500 bash scripts/gpu_measure.sh 2 3 4 5 2>&1 | tail -12
echo "== sweep (defaults)"; timeout 600 python scripts/mapping_sweep.py --algos aba,aba_walk,aba_banks --batches 4096,8192,12288,16384,65536 2>&1 | grep -v amdgpu.ids | tee gpurun_out/walk_sweep.txt
