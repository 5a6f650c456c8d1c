#!/bin/bash
# ONE measurement pass for the build that is in the tree (run on the GPU box: gpurun --timeout 1500 -- 'bash scripts/gpu_measure.sh [configs]'):
#   for every BASELINE config asked for (default "2 3 4 5")
#     1. rocprofv3 --kernel-trace --stats of `bench.py --config C`      -> profiles/r03_kernel_stats_cC.csv
#     2. rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE in separate passes, then two SQ passes; no trace domains mixed in)
#                                                                       -> profiles/r03_pmc_cC.txt and one entry of profiles/r03_pmc_traffic.json
#     3. `bench.py --config C` (the line the driver would record, now carrying roofline.traffic measured on THESE sources)
#                                                                       -> profiles/r03_bench_cC.json
# profiles/r03_pmc_traffic.json records bench.kernel_source_hash(); bench.py ignores the file when the sources have changed since.
# Everything is also copied to gpurun_out/profiles/ so that it comes back from the box; copy it from there into profiles/ and commit.
CONFIGS=${*:-2 3 4 5}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
OUT=$R/gpurun_out/measure; rm -rf $OUT; mkdir -p $OUT $R/gpurun_out/profiles
cd /tmp
for C in $CONFIGS; do
  SHORT="--config $C --no-cpu-baseline --no-pipelined --no-other-configs --steps 60 --warmup 10"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c$C -- python $R/bench.py $SHORT > $OUT/stats_c$C.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_c$C -- python $R/bench.py $SHORT > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_c$C -- python $R/bench.py $SHORT > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq1_c$C -- python $R/bench.py $SHORT > /dev/null 2>&1
  rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq2_c$C -- python $R/bench.py $SHORT > /dev/null 2>&1
done
cd $R
python scripts/summarize_measure.py $CONFIGS
for C in $CONFIGS; do
  python bench.py --config $C > profiles/r03_bench_c$C.json 2> $OUT/bench_c$C.err
  tail -c 600 profiles/r03_bench_c$C.json | head -c 0
  python - <<PY
import json
d = json.load(open("profiles/r03_bench_c$C.json"))
print("config $C:", d["value"], d["unit"], "ms/step", round(d["ms_per_step"], 4), "kernel_ms", d["roofline"]["kernel_ms"], "traffic", d["roofline"]["traffic"], "stale", d["roofline"]["traffic_stale"])
PY
done
cp profiles/r03_kernel_stats_c*.csv profiles/r03_pmc_c*.txt profiles/r03_pmc_traffic.json profiles/r03_bench_c*.json gpurun_out/profiles/ 2>/dev/null
