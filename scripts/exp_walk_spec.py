"""dynamics! through the walk kernel compiled for the mechanism (aba_walk_spec, csrc/rbd_walk.hpp; fp64): parity against the oracle, then graph-replayed µs
per launch.  Run once as is and once with RBD_TUNE=spec_walk_min_batch=1000000000 (the interpreting walk kernel) to compare."""
import os, sys
os.environ.setdefault("RBD_JIT_ASYNC", "0")  # wait for the kernels compiled per mechanism instead of starting on the interpreting ones
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import rbd_amd as rbd
import oracle

def load(name): return rbd.load_flat_model(os.path.join(ROOT, "tests", "golden", "models", name + ".json"))
names = os.environ.get("MODELS", "atlas_floating").split(",")
rng = np.random.default_rng(5)
for name in names:
    model = load(name)
    B = 300
    q = rbd.rand_configuration(model, B, rng); v = rbd.rand_velocity(model, B, rng)
    tau = rng.standard_normal((B, model.nv)); fe = rng.standard_normal((B, 6 * model.n_bodies))
    for layout in ("aos", "soa"):
        state = rbd.MechanismState(model, B, dtype=torch.float64, layout=layout)
        rbd.set_configuration_(state, q); rbd.set_velocity_(state, v)
        result = rbd.DynamicsResult(model, B, dtype=torch.float64, layout=layout)
        dev = lambda a: torch.as_tensor(a if layout == "aos" else np.ascontiguousarray(a.T), dtype=torch.float64, device="cuda")
        host = lambda t: (t if layout == "aos" else t.T).double().cpu().numpy()
        for wf in (True, False):
            rbd.dynamics_(result, state, dev(tau), dev(fe) if wf else None, algorithm="aba_walk")
            torch.cuda.synchronize()
            ref, qd = oracle.dynamics(model, q, v, tau, fe if wf else None, want_qdot=True)
            got = host(result.vd)
            print(name, layout, "fext" if wf else "no fext", "vd rel err", float(np.abs(got - ref).max() / max(1, np.abs(ref).max())),
                  "qd err", float(np.abs(host(result.qd) - qd).max()), rbd.last_kernel(state), flush=True)

model = load("atlas_floating")
for B in (16384, 65536):
    state = rbd.MechanismState(model, B, dtype=torch.float64); result = rbd.DynamicsResult(model, B, dtype=torch.float64)
    rbd.set_configuration_(state, rbd.rand_configuration(model, B, rng)); rbd.set_velocity_(state, rbd.rand_velocity(model, B, rng))
    tau = torch.rand(B, model.nv, dtype=torch.float64, device="cuda")
    f = lambda: rbd.dynamics_(result, state, tau)
    for _ in range(5): f()
    torch.cuda.synchronize()
    cap = torch.cuda.Stream(); g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(cap):
        f()
        with torch.cuda.graph(g, stream=cap):
            for _ in range(10): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g.replay(); torch.cuda.synchronize()
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    print("B", B, rbd.last_kernel(state), "us per launch", round(e0.elapsed_time(e1) * 1000 / 50, 2), flush=True)

# inverse_dynamics! through rnea_walk_spec: parity (with the per-body outputs), then timing with and without them
model = load("atlas_floating")
B = 300
q = rbd.rand_configuration(model, B, rng); v = rbd.rand_velocity(model, B, rng)
vd = rng.standard_normal((B, model.nv)); fe = rng.standard_normal((B, 6 * model.n_bodies))
os.environ.setdefault("RBD_TUNE", "spec_walk_min_batch=1")
state = rbd.MechanismState(model, B); rbd.set_configuration_(state, q); rbd.set_velocity_(state, v)
t64 = lambda a: torch.as_tensor(a, dtype=torch.float64, device="cuda")
out = torch.zeros(B, model.nv, dtype=torch.float64, device="cuda"); jw = torch.zeros(B, 6 * model.n_bodies, dtype=torch.float64, device="cuda"); acc = torch.zeros_like(jw)
rbd.inverse_dynamics_(out, state, t64(vd), t64(fe), mapping="walk", jointwrenchesout=jw, accelerations=acc)
torch.cuda.synchronize()
ref = oracle.inverse_dynamics(model, q, v, vd, fe)
print("inverse_dynamics", rbd.last_kernel(state), "tau err", float(np.abs(out.cpu().numpy() - ref).max() / max(1, np.abs(ref).max())), flush=True)
jw0, acc0 = jw.clone(), acc.clone()
os.environ["RBD_JIT"] = "0"
state0 = rbd.MechanismState(model, B); rbd.set_configuration_(state0, q); rbd.set_velocity_(state0, v)
rbd.inverse_dynamics_(out, state0, t64(vd), t64(fe), mapping="walk", jointwrenchesout=jw, accelerations=acc)
torch.cuda.synchronize()
print("against", rbd.last_kernel(state0), "jointwrenches diff", float((jw - jw0).abs().max() / jw.abs().max()), "accelerations diff", float((acc - acc0).abs().max() / acc.abs().max()), flush=True)
os.environ["RBD_JIT"] = "1"
for B in (16384, 65536):
    state = rbd.MechanismState(model, B)
    rbd.set_configuration_(state, rbd.rand_configuration(model, B, rng)); rbd.set_velocity_(state, rbd.rand_velocity(model, B, rng))
    vdd = torch.rand(B, model.nv, dtype=torch.float64, device="cuda"); out = torch.zeros_like(vdd)
    jw = torch.zeros(B, 6 * model.n_bodies, dtype=torch.float64, device="cuda"); acc = torch.zeros_like(jw)
    for label, f in (("plain (walk mapping)", lambda: rbd.inverse_dynamics_(out, state, vdd, mapping="walk")), ("plain (default)", lambda: rbd.inverse_dynamics_(out, state, vdd)),
                     ("with per-body outputs", lambda: rbd.inverse_dynamics_(out, state, vdd, jointwrenchesout=jw, accelerations=acc))):
        for _ in range(5): f()
        torch.cuda.synchronize()
        cap = torch.cuda.Stream(); g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(cap):
            f()
            with torch.cuda.graph(g, stream=cap):
                for _ in range(10): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        g.replay(); torch.cuda.synchronize()
        e0.record()
        for _ in range(5): g.replay()
        e1.record(); torch.cuda.synchronize()
        print("B", B, "inverse_dynamics!", label, rbd.last_kernel(state), "us per launch", round(e0.elapsed_time(e1) * 1000 / 50, 2), flush=True)
