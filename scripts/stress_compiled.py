"""One-off stress of the kernels compiled per mechanism at run time (csrc/rbd_spec.hpp through rbd_jit.hip): random tree topologies — chains, bushes,
branch points below branch points, fixed / prismatic / sin-cos joints, with and without a 6-dof root — through aba_spec, rnea_spec, crba_spec (+ the
sparse Cholesky and the emitter when nv is a multiple of 4) against the oracle.  Every tree costs three hiprtc compilations (seconds each)."""
import os, sys
os.environ.setdefault("RBD_JIT_ASYNC", "0")  # wait for the kernels compiled per mechanism instead of starting on the interpreting ones
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
os.environ["RBD_TUNE"] = "state_min_batch=1"
import numpy as np, torch
import rbd_amd as rbd, oracle
from test_chain_plan import random_tree
N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(77)
worst, done, skipped = {}, 0, 0
f32 = lambda a: a.astype(np.float32).astype(np.float64)
sym = lambda M: np.tril(M) + np.transpose(np.tril(M, -1), (0, 2, 1))
for trial in range(N):
    n = int(rng.integers(1, 34))
    mech = random_tree(rbd, rng, n, bool(trial % 2), float(rng.uniform(0, 1)))
    model = rbd.flatten(mech)
    if model.nv == 0:
        continue
    B = int(rng.integers(1, 200))
    r2 = np.random.default_rng(trial)
    q, v = f32(rbd.rand_configuration(model, B, r2)), f32(rbd.rand_velocity(model, B, r2))
    tau, fe, vd = f32(r2.random((B, model.nv))), f32(r2.random((B, 6 * model.n_bodies))), f32(r2.standard_normal((B, model.nv)))
    try:
        state = rbd.MechanismState(model, B, dtype=torch.float32)
    except Exception:
        skipped += 1
        continue
    res = rbd.DynamicsResult(model, B, dtype=torch.float32)
    rbd.set_configuration_(state, q); rbd.set_velocity_(state, v)
    t = lambda a: torch.as_tensor(a, dtype=torch.float32, device="cuda")
    Mr = oracle.mass_matrix(model, q); Ms = sym(Mr)
    c = oracle.dynamics_bias(model, q, v, fe)
    def upd(k, e): worst[k] = max(worst.get(k, 0.0), float(e))
    try:
        rbd.dynamics_(res, state, t(tau), t(fe), algorithm="aba_compiled")
        got = res.vd.double().cpu().numpy()
        r = np.einsum("bij,bj->bi", Ms, got) - (tau - c)
        eta = np.linalg.norm(r, axis=1) / (np.linalg.norm(Ms, axis=(1, 2)) * np.linalg.norm(got, axis=1) + np.linalg.norm(tau - c, axis=1) + 1e-300)
        upd("aba_spec backward error", eta.max()); assert eta.max() < 1e-5, (trial, n, B, eta.max())
        _, qd = oracle.dynamics(model, q, v, tau, fe, want_qdot=True)
        e = np.abs(res.qd.double().cpu().numpy() - qd).max() / max(1.0, np.abs(qd).max())
        upd("aba_spec qdot", e); assert e < 1e-5, (trial, e)
    except rbd._capi.RBDError as ex:
        assert ex.status == 3; skipped += 1
    try:
        out = torch.zeros_like(state.v)
        rbd.inverse_dynamics_(out, state, t(vd), t(fe), mapping="compiled")
        ref = oracle.inverse_dynamics(model, q, v, vd, fe)
        e = np.abs(out.double().cpu().numpy() - ref).max() / max(1.0, np.abs(ref).max())
        upd("rnea_spec", e); assert e < 1e-4, (trial, n, B, e)
    except rbd._capi.RBDError as ex:
        assert ex.status == 3; skipped += 1
    x = torch.zeros_like(state.v)
    Mout = torch.full((B, model.nv * model.nv), float("nan"), dtype=torch.float32, device="cuda")
    rbd.mass_matrix_solve_(x, state, t(tau), Mout)
    assert rbd.sync(state) == 0, (trial, rbd.last_kernel(state))
    kern = rbd.last_kernel(state)
    il = np.tril_indices(model.nv)
    got = Mout.double().cpu().numpy().reshape(B, model.nv, model.nv).transpose(0, 2, 1)
    e = np.abs(got[:, il[0], il[1]] - Mr[:, il[0], il[1]]).max() / np.abs(Mr).max()
    upd("mass matrix (" + kern.split(" ")[0] + ")", e); assert e < 1e-5, (trial, kern, e)
    xg = x.double().cpu().numpy()
    r = np.einsum("bij,bj->bi", Ms, xg) - tau
    eta = np.linalg.norm(r, axis=1) / (np.linalg.norm(Ms, axis=(1, 2)) * np.linalg.norm(xg, axis=1) + np.linalg.norm(tau, axis=1))
    upd("solve backward error (" + kern.split("+")[-1].strip().split(" ")[0] + ")", eta.max()); assert eta.max() < 1e-4, (trial, kern, eta.max())
    Mo = torch.full((B, model.nv * model.nv), float("nan"), dtype=torch.float32, device="cuda")
    rbd.mass_matrix_(Mo, state)
    got = Mo.double().cpu().numpy().reshape(B, model.nv, model.nv).transpose(0, 2, 1)
    e = np.abs(got[:, il[0], il[1]] - Mr[:, il[0], il[1]]).max() / np.abs(Mr).max()
    upd("mass_matrix! alone (" + rbd.last_kernel(state).split(" ")[0] + ")", e); assert e < 1e-5, (trial, e)
    done += 1
    print("tree", trial, "bodies", n, "nv", model.nv, "B", B, "ok", flush=True)
print(done, "trees ok,", skipped, "skips; worst:", worst)
