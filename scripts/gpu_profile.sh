#!/bin/bash
# Profiling pass: phase ablation of aba_kernel + rocprofv3 PMC counters (separate passes, no trace domains mixed in).
mkdir -p gpurun_out/prof2
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for ph in 1 2 3 4 5 0; do
  echo "-- stop_after=$ph"
  RBD_ABA_STOP_AFTER=$ph python - <<'PY' 2>&1 | tail -1 | tee -a gpurun_out/ablation.log
import os, sys, json, subprocess
ph = os.environ["RBD_ABA_STOP_AFTER"]
# parity assert must be skipped for partial kernels: run bench pieces manually
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "oracle"))
import numpy as np, torch, ctypes, time
import rbd_amd as rbd
from rigidbodydynamics_jl_amd import _capi
res = {}
for dt, tdt in (("f64", torch.float64), ("f32", torch.float32)):
  for B in (4096, 65536):
    model = rbd.load_flat_model("tests/golden/models/atlas_floating.json")
    rng = np.random.default_rng(1)
    state = rbd.MechanismState(model, B, dtype=tdt); result = rbd.DynamicsResult(model, B, dtype=tdt)
    rbd.set_configuration_(state, rbd.rand_configuration(model, B, rng)); rbd.set_velocity_(state, rbd.rand_velocity(model, B, rng))
    tau = torch.rand(B, model.nv, dtype=tdt, device="cuda")
    for _ in range(20): rbd.dynamics_(result, state, tau)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 300 if B == 4096 else 50
    e0.record()
    for _ in range(n): rbd.dynamics_(result, state, tau)
    e1.record(); torch.cuda.synchronize()
    res[f"{dt}_B{B}_us"] = round(e0.elapsed_time(e1) / n * 1e3, 2)
print(json.dumps({"stop_after": int(ph), **res}))
PY
done
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/prof2/pmc1 -- python $R/bench.py --no-cpu-baseline --steps 100 --warmup 10 > $R/gpurun_out/prof2/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $R/gpurun_out/prof2/pmc2 -- python $R/bench.py --no-cpu-baseline --steps 100 --warmup 10 > $R/gpurun_out/prof2/pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/prof2/pmc3 -- python $R/bench.py --no-cpu-baseline --steps 100 --warmup 10 > $R/gpurun_out/prof2/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE GRBM_COUNT --output-format csv -d $R/gpurun_out/prof2/pmc4 -- python $R/bench.py --no-cpu-baseline --steps 100 --warmup 10 > $R/gpurun_out/prof2/pmc4.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/prof2/pmc*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "aba_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        print(f, {k: (sum(v) / len(v), len(v)) for k, v in acc.items()})
PY
