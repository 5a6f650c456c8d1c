#!/bin/bash
# banked kernel asking for q / v / tau by offsets from the kernel arguments (same round trip as the body records): whole GPU suite, then the headline sweep
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|Error" | tail -3
for dt in f64 f32; do timeout 300 python scripts/mapping_sweep.py --algos aba_banks --batches 1024,4096,8192 --dtypes $dt 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/bank_pk.txt; done
timeout 300 python scripts/mapping_sweep.py --algos aba_banks --batches 4096 --dtypes f64 --wrenches 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/bank_pk.txt
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'])"
