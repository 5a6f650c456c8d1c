export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== async test"; timeout 600 python -m pytest tests/test_state_kernels.py -x -q -m gpu -k "first_call" 2>&1 | tail -5
for v in "" maxilp iterilp maxmem w1; do
  if [ -n "$v" ]; then export RBD_LIB=$PWD/rigidbodydynamics.jl_amd/csrc/librbd_hip_$v.so; fi
  for rep in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --no-other-configs --no-extra-legs 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('variant [$v] steps 2000:', round(d['ms_per_step']*1e3,2), 'us/step kernel', round(d['roofline']['kernel_ms']*1e3,2), 'err', d['parity_rel_err_vs_oracle'])"
  done
  timeout 300 python bench.py --no-cpu-baseline --no-other-configs --no-extra-legs --steps 20 --warmup 5 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('variant [$v] steps 20  :', round(d['ms_per_step']*1e3,2), 'us/step kernel', round(d['roofline']['kernel_ms']*1e3,2))"
done
unset RBD_LIB
echo "== measure"; bash scripts/gpu_measure.sh 2>&1 | tail -30
