#!/bin/bash
# whole GPU suite with the role-pipelined mapping in it, then the random-tree stress over every mapping
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest gpu"; timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee gpurun_out/pytest_gpu.log
echo "== stress"; timeout 600 python scripts/stress_mappings.py 2>&1 | tail -2 | cut -c1-1500
