#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
timeout 1700 python -m pytest tests -x -q -m gpu 2>&1 | tail -8
python bench.py --config 4 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['kernel'], d['parity_check'])"
} > gpurun_out/r3d.log 2>&1
cat gpurun_out/r3d.log
