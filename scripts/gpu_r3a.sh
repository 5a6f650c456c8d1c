#!/bin/bash
# round 3, call A: the rebuilt banked kernel — parity subset, headline bench for the default build and the switch variants, phase timeline, PMC
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
python -m pytest tests/test_gpu_parity.py tests/test_golden_vectors.py -x -q -m gpu -k "chains or banked or isolated or golden or simulate_banked or per_body or batch_sizes or dynamics_f64 or dynamics_f32" 2>&1 | tail -4
run() { python bench.py --no-cpu-baseline --no-pipelined $@ 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  ', '$*', ':', round(d['value']/1e6,1),'Mevals/s ms_per_step', round(d['ms_per_step']*1e3,2), 'kernel_us', round(d['roofline']['kernel_ms']*1e3,2), 'err', d.get('parity_rel_err_vs_oracle'))"; }
echo "default lib"; run --steps 2000; run --steps 20 --warmup 5; run --dtype f32 --steps 2000; run --batch 8192 --steps 500 --algorithm aba_banks; RBD_BANK_GENERIC=1 run --steps 2000
for v in w1 fklds holds all; do
  echo "variant $v"; export RBD_LIB=$R/rigidbodydynamics.jl_amd/csrc/librbd_hip_$v.so
  run --steps 2000; run --dtype f32 --steps 2000
  unset RBD_LIB
done
RBD_LIB=$R/rigidbodydynamics.jl_amd/csrc/librbd_hip_prof.so python scripts/bank_phases.py 4096 f64
cd /tmp
S="--no-cpu-baseline --no-pipelined --steps 60 --warmup 10"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/r3a_pmc1 -- python $R/bench.py $S > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $R/gpurun_out/r3a_pmc2 -- python $R/bench.py $S > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3a_stats -- python $R/bench.py $S > /dev/null 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
for d in ("r3a_pmc1", "r3a_pmc2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] in ("SQ_WAVES", "SQ_INSTS_SALU"): n[k] += 1
    for k, v in acc.items():
        if "aba" not in k: continue
        print(k, "dispatches", n[k])
        w = None
        for c, x in sorted(v.items()): print(f"   {c:24s} {x / max(n[k],1):14.1f} per launch   {x / max(n[k],1) / 1024:10.1f} per wave(1024)")
for f in glob.glob("gpurun_out/r3a_stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "aba" in r["Name"]: print(r["Name"][:70], r["Calls"], r["AverageNs"])
PY
