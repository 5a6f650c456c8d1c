#!/bin/bash
# `simulate` and plain dynamics! at 65 536 fp64 states per generated-code variant (RBD_TUNE spec_variant=<bits>): us per RK4 step / per launch.
# usage: gpurun -- 'bash scripts/gpu_sim_variants.sh 0 64 128'
R=${GRAFT_REPO_ROOT:-$(pwd)}
for V in "$@"; do
  export RBD_TUNE="spec_variant=$V"
  echo "variant $V: $(python $R/scripts/sim_prof.py 65536 f64 | tail -1)"
  python $R/bench.py --config 2 --batch 65536 --no-cpu-baseline --no-extra-legs --no-other-configs --steps 60 --warmup 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   dynamics! ms/step', round(d['ms_per_step'],5), 'kernel_ms', d['roofline'].get('kernel_ms'))"
done
