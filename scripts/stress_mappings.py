"""One-off stress: many random tree topologies, every ABA / RNEA lane mapping against the oracle (fp64)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import rbd_amd as rbd, oracle
from test_chain_plan import random_tree
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(2024)
worst = {}
skipped = {"aba_banks": 0, "aba_walk": 0}
for trial in range(N):
    n = int(rng.integers(1, 45))
    mech = random_tree(rbd, rng, n, bool(rng.integers(2)), float(rng.uniform(0, 1)))
    model = rbd.flatten(mech)
    if model.nv == 0:  # all joints Fixed: nothing to compare (tests/test_gpu_parity.py::test_mechanism_without_degrees_of_freedom)
        continue
    B = int(rng.integers(1, 200))
    r2 = np.random.default_rng(trial)
    q, v = rbd.rand_configuration(model, B, r2), rbd.rand_velocity(model, B, r2)
    tau, fe = r2.random((B, model.nv)), r2.random((B, 6 * model.n_bodies))
    vd = r2.standard_normal((B, model.nv))
    try:
        state = rbd.MechanismState(model, B)
    except Exception as e:  # more than 8 children on one body: outside the library's limits
        skipped["model"] = skipped.get("model", 0) + 1
        continue
    res = rbd.DynamicsResult(model, B)
    rbd.set_configuration_(state, q); rbd.set_velocity_(state, v)
    ref = oracle.dynamics(model, q, v, tau, fe)
    t, f = torch.as_tensor(tau).cuda(), torch.as_tensor(fe).cuda()
    for algo in ("aba_lanes", "aba_banks", "aba_walk"):
        try:
            rbd.dynamics_(res, state, t, f, algorithm=algo)
        except Exception:
            assert algo != "aba_lanes", (trial, n, B)  # the one-body-per-lane mapping takes every tree
            skipped[algo] += 1
            continue
        err = np.abs(res.vd.cpu().numpy() - ref).max() / max(1.0, np.abs(ref).max())
        worst[algo] = max(worst.get(algo, 0.0), err)
        assert err < 1e-8, (trial, algo, n, B, err)
    ref = oracle.inverse_dynamics(model, q, v, vd, fe)
    out = torch.zeros_like(t)
    for mp in ("lanes", "banks", "walk"):
        try:
            rbd.inverse_dynamics_(out, state, torch.as_tensor(vd).cuda(), f, mapping=mp)
        except Exception:
            continue
        err = np.abs(out.cpu().numpy() - ref).max() / max(1.0, np.abs(ref).max())
        worst["rnea_" + mp] = max(worst.get("rnea_" + mp, 0.0), err)
        assert err < 1e-9, (trial, mp, n, B, err)
    # the walk kernels' packed fp32 form (two states per lane) on every fourth tree
    if trial % 4 == 0:
        os.environ["RBD_TUNE"] = "walk_pair_min_batch=1"
        try:
            s32 = rbd.MechanismState(model, B, dtype=torch.float32); r32 = rbd.DynamicsResult(model, B, dtype=torch.float32)
            rbd.set_configuration_(s32, q); rbd.set_velocity_(s32, v)
            o32 = torch.zeros_like(s32.v)
            rbd.inverse_dynamics_(o32, s32, torch.as_tensor(vd, dtype=torch.float32).cuda(), torch.as_tensor(fe, dtype=torch.float32).cuda(), mapping="walk")
            err = np.abs(o32.double().cpu().numpy() - ref).max() / max(1.0, np.abs(ref).max())
            worst["rnea_walk_f32x2"] = max(worst.get("rnea_walk_f32x2", 0.0), err)
            assert err < 5e-4, (trial, "rnea_walk_f32x2", n, B, err)
            rbd.dynamics_(r32, s32, torch.as_tensor(tau, dtype=torch.float32).cuda(), torch.as_tensor(fe, dtype=torch.float32).cuda(), algorithm="aba_walk")
            back = oracle.inverse_dynamics(model, q, v, r32.vd.double().cpu().numpy(), fe)
            cb = oracle.dynamics_bias(model, q, v, fe)
            berr = (np.linalg.norm(back - tau, axis=1) / np.linalg.norm(tau - cb, axis=1)).max()
            worst["aba_walk_f32x2_backward"] = max(worst.get("aba_walk_f32x2_backward", 0.0), berr)
            assert berr < 1e-4, (trial, "aba_walk_f32x2", n, B, berr)
        except rbd._capi.RBDError:
            skipped["walk_f32x2"] = skipped.get("walk_f32x2", 0) + 1
        finally:
            os.environ.pop("RBD_TUNE", None)
    # kinematics by-products on the same tree
    A = torch.zeros(B, 6 * model.nv, dtype=torch.float64, device="cuda")
    rbd.momentum_matrix_(A, state)
    A_ref, h_ref, com_ref = oracle.momentum_matrix(model, q, v)
    err = np.abs(A.cpu().numpy().reshape(B, model.nv, 6).transpose(0, 2, 1) - A_ref).max() / max(1.0, np.abs(A_ref).max())
    worst["momentum_matrix"] = max(worst.get("momentum_matrix", 0.0), err)
    assert err < 1e-10, (trial, "momentum_matrix", err)
    hb_ref = oracle.momentum(model, q, v)[1]
    err = np.abs(rbd.momentum_rate_bias(state).cpu().numpy() - hb_ref).max() / max(1.0, np.abs(hb_ref).max())
    worst["momentum_rate_bias"] = max(worst.get("momentum_rate_bias", 0.0), err)
    assert err < 1e-9, (trial, "momentum_rate_bias", err)
    base, body = (int(x) for x in r2.choice(np.arange(-1, model.n_bodies), 2, replace=model.n_bodies < 1))
    rbd.geometric_jacobian_(A, state, base, body)
    J_ref, _ = oracle.geometric_jacobian(model, q, base, body)
    err = np.abs(A.cpu().numpy().reshape(B, model.nv, 6).transpose(0, 2, 1) - J_ref).max()
    assert err < 1e-11, (trial, "jacobian", err)
    # mass matrix (lower triangle) and fp32 ABA backward error
    rbd.mass_matrix_(res, state)
    M_ref = oracle.mass_matrix(model, q)
    Mg = res.massmatrix.cpu().numpy().reshape(B, model.nv, model.nv).transpose(0, 2, 1)
    err = np.abs(np.tril(Mg) - np.tril(M_ref)).max() / max(1.0, np.abs(M_ref).max())
    worst["mass_matrix"] = max(worst.get("mass_matrix", 0.0), err)
    assert err < 1e-10, (trial, "mass_matrix", err)
print(f"{N} random trees ok; worst relative errors {worst}; out of scope: {skipped}")
