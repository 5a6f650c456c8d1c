"""dynamics! through the kernel compiled for the mechanism (aba_spec, csrc/rbd_spec.hpp): parity against the oracle on a few mechanisms, then
graph-replayed µs per launch against the walk kernel at large batches (fp32)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import rbd_amd as rbd
import oracle

def load(name): return rbd.load_flat_model(os.path.join(ROOT, "tests", "golden", "models", name + ".json"))
models = {n: load(n) for n in ("atlas_floating", "atlas_fixed", "valkyrie_floating", "acrobot_urdf")}
models["double_pendulum"] = rbd.flatten(rbd.double_pendulum())
rng = np.random.default_rng(5)
for name, model in ([] if os.environ.get('TIMING_ONLY') else models.items()):
    B = 200
    q = rbd.rand_configuration(model, B, rng).astype(np.float32).astype(np.float64); v = rbd.rand_velocity(model, B, rng).astype(np.float32).astype(np.float64)
    tau = rng.standard_normal((B, model.nv)).astype(np.float32).astype(np.float64); fe = rng.standard_normal((B, 6 * model.n_bodies)).astype(np.float32).astype(np.float64)
    for layout in ("aos", "soa"):
        state = rbd.MechanismState(model, B, dtype=torch.float32, layout=layout)
        rbd.set_configuration_(state, q); rbd.set_velocity_(state, v)
        result = rbd.DynamicsResult(model, B, dtype=torch.float32, layout=layout)
        dev = lambda a: torch.as_tensor(a if layout == "aos" else np.ascontiguousarray(a.T), dtype=torch.float32, device="cuda")
        host = lambda t: (t if layout == "aos" else t.T).double().cpu().numpy()
        for wf in (True, False):
            try:
                rbd.dynamics_(result, state, dev(tau), dev(fe) if wf else None, algorithm="aba_compiled")
            except Exception as e:
                print(name, layout, "FAILED", e); break
            torch.cuda.synchronize()
            ref, qd = oracle.dynamics(model, q, v, tau, fe if wf else None, want_qdot=True)
            M = oracle.mass_matrix(model, q); c = oracle.dynamics_bias(model, q, v, fe if wf else None)
            Ms = np.tril(M) + np.transpose(np.tril(M, -1), (0, 2, 1))
            got = host(result.vd)
            res = np.einsum("bij,bj->bi", Ms, got) - (tau - c)
            eta = np.linalg.norm(res, axis=1) / (np.linalg.norm(Ms, axis=(1, 2)) * np.linalg.norm(got, axis=1) + np.linalg.norm(tau - c, axis=1))
            print(name, layout, "fext" if wf else "no fext", "vd rel err", float(np.abs(got - ref).max() / max(1, np.abs(ref).max())), "backward err", float(eta.max()),
                  "qd err", float(np.abs(host(result.qd) - qd).max()), rbd.last_kernel(state), flush=True)

model = models["atlas_floating"]
for B in (16384, 65536):
    state = rbd.MechanismState(model, B, dtype=torch.float32); result = rbd.DynamicsResult(model, B, dtype=torch.float32)
    rbd.set_configuration_(state, rbd.rand_configuration(model, B, rng)); rbd.set_velocity_(state, rbd.rand_velocity(model, B, rng))
    tau = torch.rand(B, model.nv, dtype=torch.float32, device="cuda")
    for alg in ("aba_compiled", "aba_walk"):
        f = lambda: rbd.dynamics_(result, state, tau, algorithm=alg)
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
        print("B", B, alg, "us per launch", round(e0.elapsed_time(e1) * 1000 / 50, 2), flush=True)
