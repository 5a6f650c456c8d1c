#!/bin/bash
# PMC counters of one ABA mapping on the bench workload: usage scripts/pmc_algo.sh <algorithm> [extra bench args]; output gpurun_out/pmc_<algorithm>.txt
ALGO=${1:-aba_tracks}; shift
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p $R/gpurun_out; rm -rf $R/gpurun_out/pmcA $R/gpurun_out/pmcB
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/pmcA -- python $R/bench.py --no-cpu-baseline --no-pipelined --steps 50 --warmup 5 --algorithm $ALGO "$@" > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM --output-format csv -d $R/gpurun_out/pmcB -- python $R/bench.py --no-cpu-baseline --no-pipelined --steps 50 --warmup 5 --algorithm $ALGO "$@" > /dev/null 2>&1
cd $R
python - <<PY | tee gpurun_out/pmc_$ALGO.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc[AB]/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "aba_" in k:
            acc[k.split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    w = sum(d["SQ_WAVES"]) / max(1, len(d["SQ_WAVES"]))
    for c, v in sorted(d.items()):
        m = sum(v) / len(v)
        print(f"  {c:24s} {m:14.0f}  per wave {m / w:12.1f}")
PY
