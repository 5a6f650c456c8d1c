# the kinematics by-products (rbd_kinematics + rbd_geometric_jacobian + rbd_momentum) at small batches: the compiled lane-per-state kernels (family 11) forced
# down (RBD_TUNE spec_kin_min_batch=1) against the library's threshold.  MODEL=randmech1 BATCHES="4096 8192" bash scripts/exp_kin_small.sh for another mechanism / other sizes
export TMPDIR=/tmp
for B in ${BATCHES:-2048 4096 8192 16384 32768}; do
  for T in default 1; do
    if [ $T = default ]; then unset RBD_TUNE; else export RBD_TUNE="spec_kin_min_batch=$T"; fi
    python bench.py --config 2 ${MODEL:+--model $MODEL} ${DTYPE:+--dtype $DTYPE} --batch $B --op-kin --no-cpu-baseline --no-extra-legs --no-other-configs --steps 60 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('B', $B, 'spec_kin_min_batch', '$T', 'us per set', round(d['ms_per_step']*1e3,2), d.get('roofline',{}).get('kernel'), 'err', d.get('parity_rel_err_vs_oracle'), {k:v for k,v in d.items() if k.startswith('each') or k=='each_us'})"
  done
done
