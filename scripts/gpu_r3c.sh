#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
python -m pytest tests -x -q -m gpu 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -6
for cfg in "4096 f64" "65536 f64" "65536 f32"; do set -- $cfg; python scripts/bench_ops.py --batch $1 --dtype $2 --reps 50 2>/dev/null | tee -a gpurun_out/r3_bench_ops.jsonl | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['batch'], d['dtype'])
for k, v in d['ops'].items(): print('   %-90s %s' % (k[:90], v))"; done
