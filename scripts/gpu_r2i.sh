#!/bin/bash
# per-kernel split of a simulate step at 65 536 states
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD; cd /tmp
for dt in f64 f32; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/simprof_$dt -- python $R/scripts/sim_prof.py 65536 $dt 2>&1 | grep "us per step"
  f=$(find /tmp/simprof_$dt -name "*kernel_stats.csv" | head -1); echo "== $dt"; cut -d, -f1-4 $f | head -8; cp $f $R/gpurun_out/sim_kernel_stats_$dt.csv
done
