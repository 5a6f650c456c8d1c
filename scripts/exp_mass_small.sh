# mass_matrix! + Cholesky solve (BASELINE configs[2]'s call) at small batches: the kernels compiled for the mechanism forced down (RBD_TUNE mass_min_batch=1)
# against the one-body-per-lane kernels (mass_min_batch = 2^40); then mass_matrix! alone and the other entry points through scripts/bench_ops.py
export TMPDIR=/tmp
for DT in ${DTYPES:-f32 f64}; do
for B in ${BATCHES:-256 512 1024 2048 4096 8192 16384}; do
  for T in 1 1099511627776; do
    export RBD_TUNE="mass_min_batch=$T"
    for V in "" "--no-emit-M" "--packed-M"; do
      [ "$DT" = f64 ] && [ -n "$V" ] && continue
      python bench.py --config 3 --dtype $DT --batch $B $V --no-cpu-baseline --no-extra-legs --no-other-configs --steps 100 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$DT B', $B, 'mass_min_batch', '$T'[:6], '$V', 'us', round(d['ms_per_step']*1e3,2), d['roofline'].get('kernel'), 'err', d.get('parity_rel_err_vs_oracle'))"
    done
  done
done
done
