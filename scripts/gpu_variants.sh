#!/bin/bash
# A/B of generated-code variants (RBD_TUNE spec_variant=<bits>, rbd_jit.hip) on one leg: bench line + rocprofv3 kernel stats per variant.
# usage: gpurun -- 'bash scripts/gpu_variants.sh c4 0 1 2 3'
LEG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
OUT=$R/gpurun_out/variants; rm -rf $OUT; mkdir -p $OUT
case $LEG in c2) A="--config 2";; c2id) A="--config 2 --op inverse_dynamics";; c3) A="--config 3";; c3noM) A="--config 3 --no-emit-M";; c3pk) A="--config 3 --packed-M";; c4) A="--config 4";; c2big) A="--config 2 --batch 65536";; c2idb) A="--config 2 --batch 65536 --op inverse_dynamics --bodies";; c5) A="--config 5";; sim64) A="--config 2 --op-sim";; sim64b) A="--config 2 --batch 65536 --op-sim";; sim32) A="--config 4 --op-sim";; kin) A="--config 2 --batch 65536 --op-kin";; esac
SHORT="$A --no-cpu-baseline --no-extra-legs --no-other-configs --steps 60 --warmup 10"
cd /tmp
for V in "$@"; do
  export RBD_TUNE="spec_variant=$V,first_use_check=0"  # (the variants are wrong by construction: the library would drop them)
  python $R/bench.py $SHORT > $OUT/bench_$V.json 2> $OUT/bench_$V.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$V -- python $R/bench.py $SHORT > $OUT/stats_$V.log 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $OUT/pmc_$V -- python $R/bench.py $SHORT > /dev/null 2>&1
done
cd $R
python - "$@" <<'PY'
import csv, glob, json, sys
out = "gpurun_out/variants"
for v in sys.argv[1:]:
    try:
        d = json.loads(open(f"{out}/bench_{v}.json").read().strip().splitlines()[-1])
        line = f"variant {v}: ms/step {d['ms_per_step']:.5f} kernel_ms {d['roofline'].get('kernel_ms')} parity {d.get('parity_rel_err_vs_oracle')}"
    except Exception as e:
        line = f"variant {v}: bench line unreadable: {e} " + open(f"{out}/bench_{v}.err").read()[-400:]
    for f in glob.glob(f"{out}/stats_{v}/**/*kernel_stats.csv", recursive=True):
        r = list(csv.DictReader(open(f)))[0]
        line += f" | {r['Name'][:30]} avg ns {r['AverageNs']}"
    tot = {}
    for f in glob.glob(f"{out}/pmc_{v}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            a = tot.setdefault((r["Kernel_Name"][:30], r["Counter_Name"]), [0.0, 0]); a[0] += float(r["Counter_Value"]); a[1] += 1
    waves = {k[0]: x[0] / x[1] for k, x in tot.items() if k[1] == "SQ_WAVES"}
    big = max(waves, key=lambda k: tot[(k, "SQ_WAVE_CYCLES")][0]) if waves else None
    if big:
        line += " | per wave: " + " ".join(f"{k[1][3:]} {x[0] / x[1] / waves[big]:.0f}" for k, x in sorted(tot.items()) if k[0] == big and k[1] != "SQ_WAVES")
    print(line)
PY
