#!/bin/bash
# chain-scheduled ABA: parity tests, then a batch sweep of both lane mappings (kernel µs from HIP events)
mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu -k "chains or chain" 2>&1 | tail -15 | tee gpurun_out/pytest_chain.log
: > gpurun_out/chain_sweep.log
for dt in f64 f32; do
for B in 512 4096 16384 65536; do
for algo in aba_lanes aba_banks aba_chains; do
steps=$((400000000 / (B * 100) + 20))
timeout 300 python bench.py --no-cpu-baseline --dtype $dt --batch $B --steps $steps --warmup 10 --algorithm $algo 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$dt B=$B $algo', round(d['value']/1e6,1),'Mevals/s kernel_us', round(d['roofline']['kernel_ms']*1e3,2), 'err', d['parity_rel_err_vs_oracle'])" | tee -a gpurun_out/chain_sweep.log
done; done; done
