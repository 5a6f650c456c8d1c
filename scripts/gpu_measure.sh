#!/bin/bash
# ONE measurement pass for the build that is in the tree (run on the GPU box: gpurun --timeout 1800 -- 'bash scripts/gpu_measure.sh [legs]').
# A LEG is one timed entry point of one BASELINE config, profiled in a run of its own (`bench.py --no-extra-legs`: every launch rocprofv3 sees is the leg —
# round 3 averaged the with-M and the M_out = NULL launches of config 3 together):
#   c2 = configs[1] dynamics!          c2id = configs[1] inverse_dynamics!      c3 = configs[2] mass_matrix! + Cholesky, M emitted
#   c3noM = configs[2], M_out = NULL   c4 = configs[3], one GPU's shard         c5 = configs[4] four-bar
#   c2big = configs[1]'s dynamics! at 65 536 fp64 states      c2idb = inverse_dynamics! with per-body outputs at 65 536 fp64 states
#   rmech = dynamics! of randmech() at 65 536 fp64 states (aba_spec_f64)
#   axf = configs[1] on the fixed-base Atlas       sim64 / sim64b / sim32 = the RK4 `simulate` step at 4096 fp64 / 65 536 fp64 / 65 536 fp32 states
#   kin / kin4k = the kinematics by-products (rbd_kinematics + rbd_geometric_jacobian + rbd_momentum) at 65 536 / 4096 fp64 states
# For every leg (default: all six)
#     1. rocprofv3 --kernel-trace --stats                                -> profiles/r06_kernel_stats_<leg>.csv
#     2. rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE in separate passes, then two SQ passes; no trace domains mixed in)
#                                                                        -> profiles/r06_pmc_<leg>.txt and one entry of profiles/r06_pmc_traffic.json
# then `bench.py` (the line the driver records, now carrying roofline.traffic measured on THESE sources) -> profiles/r06_bench.json, and the same at the
# driver's --steps 20 --warmup 5 -> profiles/r06_bench_driver_steps.json.
# profiles/r06_pmc_traffic.json records bench.kernel_source_hash(); bench.py marks the figures stale when the sources have changed since.
# Everything is also copied to gpurun_out/profiles/ so that it comes back from the box; copy it from there into profiles/ and commit.
LEGS=${*:-c2 c2id c3 c3noM c3pk c4 c5 c2big c2idb axf rmech sim64 sim64b sim32 kin kin4k}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
OUT=$R/gpurun_out/measure; rm -rf $OUT; mkdir -p $OUT $R/gpurun_out/profiles
leg_args() {
  case $1 in
    c2) echo "--config 2";; c2id) echo "--config 2 --op inverse_dynamics";; c3) echo "--config 3";; c3noM) echo "--config 3 --no-emit-M";; c3pk) echo "--config 3 --packed-M";;
    c4) echo "--config 4";; c5) echo "--config 5";;
    c2big) echo "--config 2 --batch 65536";; c2idb) echo "--config 2 --batch 65536 --op inverse_dynamics --bodies";;
    axf) echo "--config 2 --model atlas_fixed";; rmech) echo "--config 2 --model randmech1 --batch 65536";;
    sim64) echo "--config 2 --op-sim";; sim64b) echo "--config 2 --batch 65536 --op-sim";; sim32) echo "--config 4 --op-sim";;
    kin) echo "--config 2 --batch 65536 --op-kin";; kin4k) echo "--config 2 --op-kin";;
    *) echo "unknown leg $1" >&2; exit 1;;
  esac
}
cd /tmp
for LEG in $LEGS; do
  SHORT="$(leg_args $LEG) --no-cpu-baseline --no-extra-legs --no-other-configs --steps 60 --warmup 10"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$LEG -- python $R/bench.py $SHORT > $OUT/stats_$LEG.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$LEG -- python $R/bench.py $SHORT > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$LEG -- python $R/bench.py $SHORT > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq1_$LEG -- python $R/bench.py $SHORT > /dev/null 2>&1
  rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq2_$LEG -- python $R/bench.py $SHORT > /dev/null 2>&1
done
cd $R
python scripts/summarize_measure.py $LEGS
python bench.py --full-out profiles/r06_bench_full.json > profiles/r06_bench.json 2> $OUT/bench.err
python bench.py --steps 20 --warmup 5 --full-out profiles/r06_bench_driver_steps_full.json > profiles/r06_bench_driver_steps.json 2>> $OUT/bench.err
wc -c profiles/r06_bench.json profiles/r06_bench_driver_steps.json
python - <<PY
import json
for f in ("profiles/r06_bench.json", "profiles/r06_bench_driver_steps.json"):
    d = json.load(open(f))
    print(f, d["value"], d["unit"], "ms/step", round(d["ms_per_step"], 4), "kernel_ms", d["roofline"]["kernel_ms"], "traffic", d["roofline"]["traffic"])
    for k, b in d.items():
        if isinstance(b, dict) and "k_us" in b:
            print("  ", k, b)
        elif isinstance(b, str) and b.startswith("failed"):
            print("  ", k, b)
PY
cp profiles/r06_kernel_stats_*.csv profiles/r06_pmc_*.txt profiles/r06_pmc_traffic.json profiles/r06_bench*.json gpurun_out/profiles/ 2>/dev/null
