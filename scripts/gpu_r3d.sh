#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -6
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms']); print({k: (d[k]['ms_per_step'], d[k]['roofline']['kernel']) for k in ('inverse_dynamics','config3','config4_shard','config5')})"
} > gpurun_out/r3d.log 2>&1
cat gpurun_out/r3d.log
