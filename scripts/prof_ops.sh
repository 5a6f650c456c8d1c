#!/bin/bash
# per-kernel durations of every op at a given batch/dtype: rocprofv3 --kernel-trace --stats over scripts/bench_ops.py
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; rm -rf $R/gpurun_out/prof_ops
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ops -- python $R/scripts/bench_ops.py "$@" > /dev/null 2>&1
cd $R; python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/prof_ops/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if float(r["Percentage"]) > 0.3:
            print(f"{r['Name'][:70]:70s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:9.2f} pct={r['Percentage']}")
PY
