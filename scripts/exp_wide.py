"""(developer) the reference's randmech() — every tree joint type — at a large batch: the kernels compiled for the mechanism against the lane-per-body ones.
usage: python scripts/exp_wide.py [batch=65536]"""
import os, sys, json
os.environ.setdefault("RBD_JIT_ASYNC", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch, rbd_amd as rbd
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
model = rbd.flatten(rbd.randmech(np.random.default_rng(1)))


def timed(f, reps=50):
    for _ in range(3): f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); cap = torch.cuda.Stream()
    with torch.cuda.stream(cap):
        f()
        with torch.cuda.graph(g, stream=cap):
            for _ in range(reps): f()
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / reps * 1e3, 1)


res = {}
for dname, tdt in (("f32", torch.float32), ("f64", torch.float64)):
    rng = np.random.default_rng(1)
    state = rbd.MechanismState(model, B, dtype=tdt); result = rbd.DynamicsResult(model, B, dtype=tdt)
    rbd.set_configuration_(state, rbd.rand_configuration(model, B, rng)); rbd.set_velocity_(state, rbd.rand_velocity(model, B, rng))
    tau = torch.rand(B, model.nv, dtype=tdt, device="cuda"); out = torch.zeros_like(tau)
    for name, f in (("dynamics! auto", lambda: rbd.dynamics_(result, state, tau)), ("dynamics! lanes", lambda: rbd.dynamics_(result, state, tau, algorithm="aba_lanes")),
                    ("dynamics! compiled", (lambda: rbd.dynamics_(result, state, tau, algorithm="aba_compiled")) if dname == "f32" else None),
                    ("inverse_dynamics! compiled", lambda: rbd.inverse_dynamics_(out, state, tau, mapping="compiled")),
                    ("inverse_dynamics! auto", lambda: rbd.inverse_dynamics_(out, state, tau)), ("inverse_dynamics! lanes", lambda: rbd.inverse_dynamics_(out, state, tau, mapping="lanes")),
                    ("mass_matrix! + Cholesky", lambda: rbd.mass_matrix_solve_(out, state, tau, result.massmatrix))):
        if f is not None:
            res[f"{name} {dname}"] = [timed(f), rbd.last_kernel(state).split(" (")[0]]
print(json.dumps({"model": "randmech(seed 1)", "nb": model.n_bodies, "nv": model.nv, "batch": B, "us": res}))
