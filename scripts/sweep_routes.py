"""Which kernel should take which batch: dynamics! and inverse_dynamics! of one mechanism through every lane mapping that can be forced, against the library's own
choice ("auto"), over a range of batch sizes.  Flags the sizes where the choice is more than 5 % behind the best forced route.
usage: python scripts/sweep_routes.py [--model atlas_floating] [--dtype f64] [--batches 256,1024,...]"""
import argparse, os, sys
os.environ.setdefault("RBD_JIT_ASYNC", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import rbd_amd as rbd

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="atlas_floating"); ap.add_argument("--dtype", default="f64")
ap.add_argument("--batches", default="256,512,1024,2048,4096,8192,12288,16384,24576,32768,49152,65536")
ap.add_argument("--reps", type=int, default=30); ap.add_argument("--op", default="both", help="dynamics | inverse | both")
args = ap.parse_args()
tdt = torch.float64 if args.dtype == "f64" else torch.float32
if args.model.startswith("randmech"):
    model = rbd.flatten(rbd.randmech(np.random.default_rng(int(args.model[8:] or 1))))
elif os.path.exists(os.path.join(ROOT, "tests", "golden", "models", args.model + ".json")):
    model = rbd.load_flat_model(os.path.join(ROOT, "tests", "golden", "models", args.model + ".json"))
else:  # one of the test suite's mechanisms (tests/conftest.py build_models)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import build_models
    model = build_models(rbd)[args.model]


def timed(f, reps):
    try:
        for _ in range(3): f()
    except rbd._capi.RBDError as e:
        return None
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for op, routes in (("dynamics!", ["aba", "aba_lanes", "aba_banks", "aba_walk", "aba_compiled"]), ("inverse_dynamics!", ["auto", "lanes", "banks", "walk", "compiled"])):
    if args.op != "both" and not op.startswith(args.op):
        continue
    print(f"== {op} {args.model} {args.dtype}: us per call by route (first column: the library's choice)")
    print("       B  " + "  ".join(f"{r:>13s}" for r in routes) + "   chosen kernel")
    for B in [int(b) for b in args.batches.split(",")]:
        rng = np.random.default_rng(1)
        state = rbd.MechanismState(model, B, dtype=tdt); result = rbd.DynamicsResult(model, B, dtype=tdt)
        rbd.set_configuration_(state, rbd.rand_configuration(model, B, rng)); rbd.set_velocity_(state, rbd.rand_velocity(model, B, rng))
        tau = torch.rand(B, model.nv, dtype=tdt, device="cuda"); out = torch.zeros_like(tau)
        row, chosen = [], ""
        for r in routes:
            f = (lambda: rbd.dynamics_(result, state, tau, algorithm=r)) if op == "dynamics!" else (lambda: rbd.inverse_dynamics_(out, state, tau, mapping=r))
            row.append(timed(f, args.reps))
            if r in ("aba", "auto"): chosen = rbd.last_kernel(state)[:44]
        best = min(x for x in row[1:] if x is not None) if any(x is not None for x in row[1:]) else None
        flag = "  <-- %.0f %% behind %s" % (100 * (row[0] / best - 1), routes[1 + row[1:].index(best)]) if best and row[0] > 1.05 * best else ""
        print(f"{B:8d}  " + "  ".join(f"{x:13.1f}" if x is not None else f"{'-':>13s}" for x in row) + "   " + chosen + flag)
