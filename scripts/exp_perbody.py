"""inverse_dynamics! with and without the per-body outputs (jointwrenches, accelerations), both layouts: µs per launch (graph-replayed).
usage: python scripts/exp_perbody.py [--model atlas_floating]"""
import argparse, json, os, sys
os.environ.setdefault("RBD_JIT_ASYNC", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import rbd_amd as rbd

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="atlas_floating"); ap.add_argument("--reps", type=int, default=100)
ap.add_argument("--cases", default="f32:65536,f64:65536,f64:4096")
args = ap.parse_args()
model = rbd.load_flat_model(os.path.join(ROOT, "tests", "golden", "models", args.model + ".json"))


def timed(f):
    for _ in range(3): f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); cap = torch.cuda.Stream()
    with torch.cuda.stream(cap):
        f()
        with torch.cuda.graph(g, stream=cap):
            for _ in range(args.reps): f()
    torch.cuda.synchronize()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / args.reps * 1e3, 2)


for case in args.cases.split(","):
    dt, B = case.split(":"); B = int(B)
    tdt = torch.float64 if dt == "f64" else torch.float32
    for layout in ("aos", "soa"):
        rng = np.random.default_rng(1)
        state = rbd.MechanismState(model, B, dtype=tdt, layout=layout)
        rbd.set_configuration_(state, rbd.rand_configuration(model, B, rng)); rbd.set_velocity_(state, rbd.rand_velocity(model, B, rng))
        shp = (lambda n: (B, n)) if layout == "aos" else (lambda n: (n, B))
        vd = torch.rand(shp(model.nv), dtype=tdt, device="cuda"); out = torch.zeros_like(vd)
        fe = torch.rand(shp(6 * model.n_bodies), dtype=tdt, device="cuda"); jw = torch.zeros_like(fe); acc = torch.zeros_like(fe)
        res = {}
        for name, kw in (("plain", {}), ("fext", dict(externalwrenches=fe)), ("jw", dict(jointwrenchesout=jw)), ("acc", dict(accelerations=acc)),
                         ("jw+acc", dict(jointwrenchesout=jw, accelerations=acc)), ("fext+jw+acc", dict(externalwrenches=fe, jointwrenchesout=jw, accelerations=acc))):
            res[name] = timed(lambda: rbd.inverse_dynamics_(out, state, vd, **kw))
            res[name + " kernel"] = rbd.last_kernel(state).split(" (")[0]
        print(json.dumps({"case": case, "layout": layout, "us": res}), flush=True)
