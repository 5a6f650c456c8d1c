#!/bin/bash
# tiny loop: headline + a few sizes, optionally a test subset (TESTS="-k banks")
mkdir -p gpurun_out; : > gpurun_out/mini.log
if [ -n "$TESTS" ]; then timeout 600 python -m pytest tests -x -q -m gpu $TESTS 2>&1 | tail -3 | tee -a gpurun_out/mini.log; fi
for extra in "" "--dtype f32" "--batch 512" "--batch 65536 --steps 200" "--algorithm aba_lanes" ${EXTRA:+"$EXTRA"}; do
timeout 300 python bench.py --no-cpu-baseline $extra 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$extra]', d['roofline']['kernel'], round(d['value']/1e6,1),'Mevals/s kernel_us', round(d['roofline']['kernel_ms']*1e3,2), 'err', d['parity_rel_err_vs_oracle'])" | tee -a gpurun_out/mini.log
done
