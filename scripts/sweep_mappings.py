#!/usr/bin/env python3
"""Launch time of `dynamics!` per lane mapping and batch size (Atlas floating): the data behind run_aba's thresholds.
usage: python scripts/sweep_mappings.py [--batches 512,1024,...] [--dtypes f64,f32] [--algos aba_lanes,aba_banks,aba_walk,aba] [--reps 300]"""
import argparse, os, sys
os.environ.setdefault("RBD_JIT_ASYNC", "0")  # wait for the kernels compiled per mechanism instead of starting on the interpreting ones
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rbd_amd as rbd
from rbd_amd import flatio

ap = argparse.ArgumentParser()
ap.add_argument("--batches", default="512,1024,2048,3072,4096,6144,8192,12288,16384")
ap.add_argument("--dtypes", default="f64,f32")
ap.add_argument("--algos", default="aba_lanes,aba_banks,aba_walk,aba")
ap.add_argument("--reps", type=int, default=300)
ap.add_argument("--model", default="atlas_floating")
args = ap.parse_args()
model = flatio.load_flat_model(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "models", args.model + ".json"))
print(f"# {args.model}: us per launch (HIP events over {args.reps} launches)")
for dt in args.dtypes.split(","):
    tdt = torch.float64 if dt == "f64" else torch.float32
    print(f"{dt:4s} {'B':>7s} " + " ".join(f"{a:>10s}" for a in args.algos.split(",")) + "   kernel picked by 'aba'")
    for B in [int(x) for x in args.batches.split(",")]:
        state = rbd.MechanismState(model, B, dtype=tdt)
        rbd.rand_(state, seed=1)
        res = rbd.DynamicsResult(model, B, dtype=tdt)
        tau = torch.rand(B, model.nv, dtype=tdt, device="cuda")
        row = []
        for a in args.algos.split(","):
            try:
                for _ in range(20):
                    rbd.dynamics_(res, state, tau, algorithm=a)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.reps):
                    rbd.dynamics_(res, state, tau, algorithm=a)
                e1.record()
                torch.cuda.synchronize()
                row.append(f"{e0.elapsed_time(e1) / args.reps * 1e3:10.2f}")
            except Exception as ex:
                row.append(f"{'n/a':>10s}")
        print(f"{'':4s} {B:7d} " + " ".join(row) + f"   {rbd.last_kernel(state)}")
