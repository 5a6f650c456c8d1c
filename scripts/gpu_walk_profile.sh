#!/bin/bash
# rocprofv3 kernel statistics and SQ counters of the compiled walk kernels beside the interpreting ones (fp64, 65 536 Atlas states) -> profiles/r03_walk_spec_*
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
OUT=$R/gpurun_out/walkprof; rm -rf $OUT; mkdir -p $OUT $R/gpurun_out/profiles
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $R/scripts/walk_spec_prof.py > $OUT/stats.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/sq1 -- python $R/scripts/walk_spec_prof.py > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/sq2 -- python $R/scripts/walk_spec_prof.py > /dev/null 2>&1
cd $R
python - <<'PY'
import csv, glob, os, collections
out = "gpurun_out/walkprof"
stats = glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True)
rows = []
for f in stats:
    for r in csv.DictReader(open(f)):
        if "walk" in r["Name"]: rows.append(r)
with open("gpurun_out/profiles/r03_walk_spec_kernel_stats.csv", "w") as g:
    if rows:
        w = csv.DictWriter(g, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
for r in rows: print(r["Name"][:60], "calls", r["Calls"], "avg ns", r["AverageNs"])
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for d in ("sq1", "sq2"):
    for f in glob.glob(out + "/" + d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "walk" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
with open("gpurun_out/profiles/r03_walk_spec_pmc.txt", "w") as g:
    g.write("SQ counters per launch (mean over the launches of scripts/walk_spec_prof.py; fp64, 65 536 Atlas states = 1024 workgroups of 4 wavefronts), and per wavefront\n")
    for k in sorted(acc):
        waves = acc[k]["SQ_WAVES"] / max(1, n[k]["SQ_WAVES"])
        g.write("\n" + k[:100] + "\n")
        for c in sorted(acc[k]):
            m = acc[k][c] / n[k][c]
            g.write("  %-24s %14.0f   per wavefront %10.1f\n" % (c, m, m / waves if waves else 0))
print(open("gpurun_out/profiles/r03_walk_spec_pmc.txt").read()[:3000])
PY
