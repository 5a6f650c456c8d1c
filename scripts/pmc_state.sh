#!/bin/bash
# PMC counters of the one-lane-per-state kernels: usage scripts/pmc_state.sh [B] [dtype]; output gpurun_out/pmc_state.txt
B=${1:-4096}; DT=${2:-f32}
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p $R/gpurun_out; rm -rf $R/gpurun_out/pmcS1 $R/gpurun_out/pmcS2 $R/gpurun_out/pmcS3
cat > /tmp/run_state.py <<PY
import os, sys
sys.path.insert(0, "$R")
import numpy as np, torch
import rbd_amd as rbd
tdt = torch.float32 if "$DT" == "f32" else torch.float64
model = rbd.load_flat_model("$R/tests/golden/models/atlas_floating.json")
rng = np.random.default_rng(1)
B = $B
state = rbd.MechanismState(model, B, dtype=tdt, layout="soa"); result = rbd.DynamicsResult(model, B, dtype=tdt, layout="soa")
rbd.set_configuration_(state, rbd.rand_configuration(model, B, rng)); rbd.set_velocity_(state, rbd.rand_velocity(model, B, rng))
for _ in range(10):
    rbd.dynamics_bias_(result, state); rbd.mass_matrix_(result, state)
torch.cuda.synchronize()
PY
cd /tmp
export RBD_STATE_MIN_BATCH=1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/pmcS1 -- python /tmp/run_state.py > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $R/gpurun_out/pmcS2 -- python /tmp/run_state.py > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_FLAT SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_IFETCH SQ_INSTS_VSKIPPED --output-format csv -d $R/gpurun_out/pmcS3 -- python /tmp/run_state.py > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pmcS4 -- python /tmp/run_state.py > /dev/null 2>&1
cd $R
python - <<PY | tee gpurun_out/pmc_state.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmcS[123]/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "_state_kernel" in k:
            acc[k.split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    w = sum(d["SQ_WAVES"]) / max(1, len(d["SQ_WAVES"]))
    for c, v in sorted(d.items()):
        m = sum(v) / len(v)
        print(f"  {c:24s} {m:14.0f}  per wave {m / w:12.1f}")
for f in glob.glob("gpurun_out/pmcS4/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "state_kernel" in r["Name"]: print(r["Name"][:60], r["Calls"], r["AverageNs"])
PY
