#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
show='import json,sys
d=json.loads(sys.stdin.read()); print(d["value"], "ms/step", d["ms_per_step"], "kernel_ms", d["roofline"]["kernel_ms"], d["steps"], d["warmup"])'
echo "config 4 default"; python bench.py --config 4 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$show"
echo "config 4 40/8"; python bench.py --config 4 --no-cpu-baseline --steps 40 --warmup 8 2>/dev/null | tail -1 | python -c "$show"
echo "config 4 40/8 walk"; RBD_JIT=0 python bench.py --config 4 --no-cpu-baseline --steps 40 --warmup 8 2>/dev/null | tail -1 | python -c "$show"
echo "config 3 40/8"; python bench.py --config 3 --no-cpu-baseline --steps 40 --warmup 8 2>/dev/null | tail -1 | python -c "$show"
echo "config 3 40/8 nojit"; RBD_JIT=0 python bench.py --config 3 --no-cpu-baseline --steps 40 --warmup 8 2>/dev/null | tail -1 | python -c "$show"
echo "headline line"; python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms']); print({k: (d[k]['ms_per_step'], d[k]['roofline']['kernel_ms']) for k in ('inverse_dynamics','config3','config4_shard','config5')})"
} > gpurun_out/r3d.log 2>&1
cat gpurun_out/r3d.log
