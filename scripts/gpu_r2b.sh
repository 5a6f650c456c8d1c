#!/bin/bash
# round-2 call B: aba_walk_kernel — parity, mapping sweep, in-kernel phase timeline, PMC of a launch
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
echo "== pytest gpu (mappings)"; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden_vectors.py -x -q -m gpu -k "chains or isolated or vectors or walk" 2>&1 | tail -8 | tee gpurun_out/pytest_walk.log
echo "== sweep"; timeout 900 python scripts/mapping_sweep.py --algos aba_walk,aba_banks,aba_tracks --batches 512,4096,8192,16384,32768,65536,262144 2>&1 | tee gpurun_out/walk_sweep.txt
echo "== sweep f32, two states per lane at every size"; RBD_WALK_PAIR_MIN_BATCH=1 timeout 600 python scripts/mapping_sweep.py --algos aba_walk --dtypes f32 --batches 4096,8192,16384,32768,65536,262144 2>&1 | tee gpurun_out/walk_sweep_pair.txt
echo "== sweep wrenches"; timeout 600 python scripts/mapping_sweep.py --algos aba_walk,aba_banks --batches 4096,65536 --wrenches 2>&1 | tee gpurun_out/walk_sweep_wrenches.txt
echo "== phases"; RBD_LIB=$R/rigidbodydynamics.jl_amd/csrc/librbd_hip_prof.so timeout 300 python scripts/walk_phases.py 2>&1 | tee gpurun_out/walk_phases.txt
echo "== pmc"
cd /tmp
CMD="python $R/bench.py --algorithm aba_walk --no-cpu-baseline --no-pipelined --steps 60 --warmup 10"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/walk_stats -- $CMD > $R/gpurun_out/walk_stats.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/walk_pmc1 -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $R/gpurun_out/walk_pmc2 -- $CMD > /dev/null 2>&1
cd $R
tail -3 gpurun_out/walk_stats.log
find gpurun_out/walk_stats -name "*kernel_stats.csv" | head -1 | xargs -r head -5
python - <<'PY'
import csv, glob, collections
for d in ("gpurun_out/walk_pmc1", "gpurun_out/walk_pmc2"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            if "aba_walk" in r["Kernel_Name"]:
                a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
        for k, (v, n) in sorted(acc.items()):
            print(f"  {k:28s} per launch {v / n:14.1f}")
PY
