#!/bin/bash
# A/B: default lib vs variants. usage: gpu_ab.sh
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -3
run() { python bench.py --no-cpu-baseline $@ 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  ', '$*', ':', round(d['value']/1e6,1),'Mevals/s kernel_us', round(d['roofline']['kernel_ms']*1e3,2), 'err', d['parity_rel_err_vs_oracle'])"; }
echo "default lib"; run; run --dtype f32; run --batch 65536 --steps 200; run --dtype f32 --batch 65536 --steps 200
for v in $@; do
  echo "variant $v"; export RBD_LIB=$PWD/rigidbodydynamics.jl_amd/csrc/librbd_hip_$v.so
  run; run --dtype f32; run --batch 65536 --steps 200; run --dtype f32 --batch 65536 --steps 200
  unset RBD_LIB
done
