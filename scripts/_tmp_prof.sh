cd /tmp; export TMPDIR=/tmp
for L in aos soa; do
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/st_$L -- python $GRAFT_REPO_ROOT/bench.py --config 3 --layout $L --no-cpu-baseline --steps 40 --warmup 5 > /dev/null 2>&1
python3 - <<PY
import csv,glob
for f in glob.glob("$GRAFT_REPO_ROOT/gpurun_out/st_$L/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if int(r["Calls"])>=40: print("$L", r["Name"][:70], r["Calls"], r["AverageNs"])
PY
done
