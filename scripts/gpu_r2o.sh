#!/bin/bash
# rocprofv3 kernel stats of the fp64 routes through the new tile Cholesky (4096 and 65 536 states)
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD; cd /tmp
for B in 4096 65536; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$B -- python $R/scripts/bench_ops.py --dtype f64 --batch $B --only Cholesky --reps 50 > /dev/null 2>&1
  cp $(find /tmp/st_$B -name "*kernel_stats.csv" | head -1) $R/gpurun_out/st_chol64_B$B.csv
done
