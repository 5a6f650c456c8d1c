#!/bin/bash
# quick loop: bank parity subset + headline + phase timeline (+ variants given as arguments)
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
python -m pytest tests/test_gpu_parity.py tests/test_golden_vectors.py -x -q -m gpu -k "chains or banked or isolated or golden or simulate_banked or per_body or batch_sizes or dynamics_f64 or dynamics_f32" 2>&1 | tail -3
run() { python bench.py --no-cpu-baseline --no-pipelined $@ 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  ', '$*', ':', round(d['value']/1e6,1),'Mevals/s ms_per_step', round(d['ms_per_step']*1e3,2), 'kernel_us', round(d['roofline']['kernel_ms']*1e3,2), 'err', d.get('parity_rel_err_vs_oracle'))"; }
echo "default lib"; run --steps 2000; run --steps 20 --warmup 5; run --dtype f32 --steps 2000; RBD_BANK_GENERIC=1 run --steps 2000
for v in $@; do
  echo "variant $v"; export RBD_LIB=$R/rigidbodydynamics.jl_amd/csrc/librbd_hip_$v.so
  run --steps 2000; run --dtype f32 --steps 2000
  unset RBD_LIB
done
RBD_LIB=$R/rigidbodydynamics.jl_amd/csrc/librbd_hip_prof.so python scripts/bank_phases.py 4096 f64 2>&1 | grep -v amdgpu.ids
