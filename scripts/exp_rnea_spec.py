"""inverse_dynamics! / dynamics_bias! through the kernel compiled for the mechanism (rnea_spec, csrc/rbd_spec.hpp): parity against the oracle, then
graph-replayed µs per launch against the walk kernel at large batches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import rbd_amd as rbd
import oracle

def load(name): return rbd.load_flat_model(os.path.join(ROOT, "tests", "golden", "models", name + ".json"))
models = {n: load(n) for n in ("atlas_floating", "atlas_fixed", "valkyrie_floating", "acrobot_urdf")}
models["double_pendulum"] = rbd.flatten(rbd.double_pendulum())
rng = np.random.default_rng(5)
for name, model in ([] if os.environ.get("TIMING_ONLY") else models.items()):
    for dt, nd in ((torch.float64, np.float64), (torch.float32, np.float32)):
        B = 150
        cast = lambda a: a.astype(nd).astype(np.float64)
        q = cast(rbd.rand_configuration(model, B, rng)); v = cast(rbd.rand_velocity(model, B, rng))
        vd = cast(rng.standard_normal((B, model.nv))); fe = cast(rng.standard_normal((B, 6 * model.n_bodies)))
        for layout in ("aos", "soa"):
            state = rbd.MechanismState(model, B, dtype=dt, layout=layout)
            rbd.set_configuration_(state, q); rbd.set_velocity_(state, v)
            dev = lambda a: torch.as_tensor(a if layout == "aos" else np.ascontiguousarray(a.T), dtype=dt, device="cuda")
            host = lambda t: (t if layout == "aos" else t.T).double().cpu().numpy()
            out = torch.zeros_like(state.v)
            try:
                rbd.inverse_dynamics_(out, state, dev(vd), dev(fe), mapping="compiled")
            except Exception as e:
                print(name, dt, layout, "FAILED", e); continue
            torch.cuda.synchronize()
            ref = oracle.inverse_dynamics(model, q, v, vd, fe)
            e1 = np.abs(host(out) - ref).max() / max(1, np.abs(ref).max())
            rbd.dynamics_bias_(out, state, mapping="compiled")
            torch.cuda.synchronize()
            ref = oracle.dynamics_bias(model, q, v, None)
            e2 = np.abs(host(out) - ref).max() / max(1, np.abs(ref).max())
            print(name, str(dt).split(".")[1], layout, "inverse_dynamics rel err", float(e1), "bias rel err", float(e2), rbd.last_kernel(state), flush=True)

model = models["atlas_floating"]
for dt in (torch.float32, torch.float64):
    for B in (16384, 32768, 65536):
        state = rbd.MechanismState(model, B, dtype=dt)
        rbd.set_configuration_(state, rbd.rand_configuration(model, B, rng)); rbd.set_velocity_(state, rbd.rand_velocity(model, B, rng))
        vd = torch.rand(B, model.nv, dtype=dt, device="cuda"); out = torch.zeros_like(vd)
        for mp in ("compiled", "walk"):
            f = lambda: rbd.inverse_dynamics_(out, state, vd, None, mapping=mp)
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
            print(str(dt).split(".")[1], "B", B, mp, "us per launch", round(e0.elapsed_time(e1) * 1000 / 50, 2), flush=True)
