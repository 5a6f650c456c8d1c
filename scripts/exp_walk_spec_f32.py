"""fp32 dynamics! / inverse_dynamics! through the walk kernels compiled for the mechanism (one and two states per lane; RBD_SPEC_WALK_F32=1) against the interpreting
walk kernels and aba_spec: backward error on a small batch, then graph-replayed µs per launch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import rbd_amd as rbd
import oracle
model = rbd.load_flat_model(os.path.join(ROOT, "tests", "golden", "models", "atlas_floating.json"))
rng = np.random.default_rng(5)
f32 = lambda a: a.astype(np.float32).astype(np.float64)
sym = lambda M: np.tril(M) + np.transpose(np.tril(M, -1), (0, 2, 1))
def timed(f):
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
    return round(e0.elapsed_time(e1) * 1000 / 50, 2)
for B in (16384, 32768, 65536):
    q, v = f32(rbd.rand_configuration(model, B, rng)), f32(rbd.rand_velocity(model, B, rng))
    tau = f32(rng.random((B, model.nv))); fe = f32(rng.standard_normal((B, 6 * model.n_bodies)))
    state = rbd.MechanismState(model, B, dtype=torch.float32); result = rbd.DynamicsResult(model, B, dtype=torch.float32)
    rbd.set_configuration_(state, q); rbd.set_velocity_(state, v)
    t = lambda a: torch.as_tensor(a, dtype=torch.float32, device="cuda")
    tt, ff = t(tau), t(fe)
    for alg in ("aba_walk", "aba"):
        rbd.dynamics_(result, state, tt, ff, algorithm=alg)
        torch.cuda.synchronize()
        kern = rbd.last_kernel(state)
        n = 512
        got = result.vd[:n].double().cpu().numpy()
        M = sym(oracle.mass_matrix(model, q[:n])); c = oracle.dynamics_bias(model, q[:n], v[:n], fe[:n])
        r = np.einsum("bij,bj->bi", M, got) - (tau[:n] - c)
        eta = np.linalg.norm(r, axis=1) / (np.linalg.norm(M, axis=(1, 2)) * np.linalg.norm(got, axis=1) + np.linalg.norm(tau[:n] - c, axis=1))
        us = timed(lambda: rbd.dynamics_(result, state, tt, algorithm=alg))
        print("B", B, "dynamics!", alg, "|", kern, "| backward error", float(eta.max()), "| us per launch", us, flush=True)
    out = torch.zeros_like(tt)
    for mapping in ("walk", "auto"):
        rbd.inverse_dynamics_(out, state, tt, ff, mapping=mapping)
        torch.cuda.synchronize()
        kern = rbd.last_kernel(state)
        ref = oracle.inverse_dynamics(model, q[:512], v[:512], tau[:512], fe[:512])
        e = float(np.abs(out[:512].double().cpu().numpy() - ref).max() / np.abs(ref).max())
        us = timed(lambda: rbd.inverse_dynamics_(out, state, tt, mapping=mapping))
        print("B", B, "inverse_dynamics!", mapping, "|", kern, "| rel err", e, "| us per launch", us, flush=True)
