#!/bin/bash
# One development iteration on the GPU box: targeted parity tests, then legs timed and counted (gpurun -- 'bash scripts/gpu_iter.sh "<pytest -k expr>" <legs...>').
K="$1"; shift
LEGS=${*:-c4}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
OUT=$R/gpurun_out/iter; rm -rf $OUT; mkdir -p $OUT
if [ -n "$K" ]; then timeout 900 python -m pytest tests -x -q -m gpu -k "$K" 2>&1 | tail -8 | tee $OUT/pytest.log; fi
leg_args() {
  case $1 in
    c2) echo "--config 2";; c2id) echo "--config 2 --op inverse_dynamics";; c3) echo "--config 3";; c3noM) echo "--config 3 --no-emit-M";; c3pk) echo "--config 3 --packed-M";;
    c4) echo "--config 4";; c5) echo "--config 5";; c2big) echo "--config 2 --batch 65536";; c2idb) echo "--config 2 --batch 65536 --op inverse_dynamics --bodies";;
    axf) echo "--config 2 --model atlas_fixed";; rmech) echo "--config 2 --model randmech1 --batch 65536";; sim64) echo "--config 2 --op-sim";; sim64b) echo "--config 2 --batch 65536 --op-sim";; sim32) echo "--config 4 --op-sim";;
    kin) echo "--config 2 --batch 65536 --op-kin";; kin4k) echo "--config 2 --op-kin";;
    *) echo "unknown leg $1" >&2; exit 1;;
  esac
}
cd /tmp
for LEG in $LEGS; do
  SHORT="$(leg_args $LEG) --no-cpu-baseline --no-extra-legs --no-other-configs --steps 60 --warmup 10"
  python $R/bench.py $SHORT > $OUT/bench_$LEG.json 2> $OUT/bench_$LEG.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$LEG -- python $R/bench.py $SHORT > $OUT/stats_$LEG.log 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq1_$LEG -- python $R/bench.py $SHORT > /dev/null 2>&1
  rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq2_$LEG -- python $R/bench.py $SHORT > /dev/null 2>&1
done
cd $R
python - $LEGS <<'PY'
import csv, glob, json, os, sys
out = "gpurun_out/iter"
for leg in sys.argv[1:]:
    try:
        d = json.loads(open(f"{out}/bench_{leg}.json").read().strip().splitlines()[-1])
        print(leg, "value", d["value"], "ms/step", d["ms_per_step"], "kernel_ms", d["roofline"].get("kernel_ms"), "parity", d.get("parity_rel_err_vs_oracle"), d.get("parity_check"))
    except Exception as e:
        print(leg, "bench line unreadable:", e, open(f"{out}/bench_{leg}.err").read()[-600:])
    for f in glob.glob(f"{out}/stats_{leg}/**/*kernel_stats.csv", recursive=True):
        for r in list(csv.DictReader(open(f)))[:4]:
            print("   ", r["Name"][:60], "calls", r["Calls"], "avg ns", r["AverageNs"])
    tot = {}
    for p in ("sq1", "sq2"):
        for f in glob.glob(f"{out}/pmc_{p}_{leg}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = (r["Kernel_Name"][:40], r["Counter_Name"])
                a = tot.setdefault(k, [0.0, 0])
                a[0] += float(r["Counter_Value"]); a[1] += 1
    waves = {k[0]: v[0] / v[1] for k, v in tot.items() if k[1] == "SQ_WAVES"}
    for k in sorted(tot):
        v = tot[k]
        if waves.get(k[0], 0) >= 256: print(f"    {k[0]:40s} {k[1]:24s} per launch {v[0] / v[1]:14.1f} per wave {v[0] / v[1] / waves[k[0]]:10.1f}")
PY
