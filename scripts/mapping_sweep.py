"""Kernel time of every lane mapping of the fused ABA (dynamics!) over batch sizes and dtypes, graph-replayed; parity of each against
the oracle on a sample.  usage: python scripts/mapping_sweep.py [--model atlas_floating] [--batches 512,4096,16384,65536]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import rbd_amd as rbd
import oracle

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="atlas_floating"); ap.add_argument("--batches", default="512,4096,16384,65536")
ap.add_argument("--algos", default="aba_tracks,aba_lanes,aba_banks,aba_chains"); ap.add_argument("--dtypes", default="f64,f32")
ap.add_argument("--reps", type=int, default=100); ap.add_argument("--wrenches", action="store_true")
args = ap.parse_args()
model = rbd.load_flat_model(os.path.join(ROOT, "tests", "golden", "models", args.model + ".json"))
for dt in args.dtypes.split(","):
    tdt = torch.float64 if dt == "f64" else torch.float32
    for B in [int(x) for x in args.batches.split(",")]:
        rng = np.random.default_rng(1)
        q, v, tau = rbd.rand_configuration(model, B, rng), rbd.rand_velocity(model, B, rng), rng.random((B, model.nv))
        fe = rng.random((B, 6 * model.n_bodies)) if args.wrenches else None
        state = rbd.MechanismState(model, B, dtype=tdt); result = rbd.DynamicsResult(model, B, dtype=tdt)
        rbd.set_configuration_(state, q); rbd.set_velocity_(state, v)
        d_tau = torch.as_tensor(tau, dtype=tdt).cuda(); d_fe = None if fe is None else torch.as_tensor(fe, dtype=tdt).cuda()
        n = min(B, 256)
        cast = lambda a: None if a is None else a[:n].astype(np.float32 if dt == "f32" else np.float64).astype(np.float64)
        ref = oracle.dynamics(model, cast(q), cast(v), cast(tau), cast(fe))
        for algo in args.algos.split(","):
            try:
                f = lambda: rbd.dynamics_(result, state, d_tau, d_fe, algorithm=algo)
                for _ in range(3): f()
                torch.cuda.synchronize()
                err = float(np.abs(result.vd[:n].double().cpu().numpy() - ref).max() / max(1.0, np.abs(ref).max()))
                g = torch.cuda.CUDAGraph(); cap = torch.cuda.Stream()
                with torch.cuda.stream(cap):
                    f()
                    with torch.cuda.graph(g, stream=cap):
                        for _ in range(args.reps): f()
                torch.cuda.synchronize(); g.replay(); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
                us = e0.elapsed_time(e1) / args.reps * 1e3
                print(f"{dt} B={B} {algo} {B / us:.1f} Mevals/s kernel_us {us:.2f} err {err:.3e}", flush=True)
            except Exception as e:
                print(f"{dt} B={B} {algo} failed: {type(e).__name__}: {str(e)[:120]}", flush=True)
