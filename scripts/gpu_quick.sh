#!/bin/bash
# quick loop: GPU parity tests + headline bench + a few variants
mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/pytest_gpu.log
timeout 600 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('f64 B4096', round(d['value']/1e6,1),'Mevals/s kernel_us', round(d['roofline']['kernel_ms']*1e3,2), 'err', d['parity_rel_err_vs_oracle'])" | tee gpurun_out/quick.log
for extra in "--dtype f32" "--batch 65536 --steps 200" "--dtype f32 --batch 65536 --steps 200" "--batch 512 --steps 2000" "--graph"; do
timeout 600 python bench.py --no-cpu-baseline $extra 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$extra', round(d['value']/1e6,1),'Mevals/s kernel_us', round(d['roofline']['kernel_ms']*1e3,2), 'err', d['parity_rel_err_vs_oracle'])" | tee -a gpurun_out/quick.log
done
