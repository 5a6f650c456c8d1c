"""Stress of the kernels compiled per mechanism ON THE HOST (tests/emu/spec_emu.py: the generated program compiled as plain C++, one lane at a time) over random
trees of every tree joint type, drawn so that sibling subtrees of the same shape — which the fp32 programs walk in lockstep (rbd_jit.hip: merge_limbs) — occur
often: dynamics! (fp32, with and without external wrenches, q̇), inverse_dynamics! with the per-body outputs and dynamics_bias! (fp32, fp64), mass_matrix!
(fp32, fp64, and the permuted staging triangle), against the oracle.  No GPU needed.
usage: python scripts/stress_spec_emu.py [N=40]"""
import os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
import numpy as np, torch
import rbd_amd as rbd, oracle, spec_emu
oracle.build()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(20260927)
ONE = ["Revolute", "Prismatic", "Fixed", "SinCosRevolute"]
ANY = ONE + ["Planar", "QuaternionSpherical", "QuaternionFloating"]


def rand_spec(depth, budget):
    """a random nested (joint, [children]) spec; with probability 1/2 a body's children come in copies of one shape (limbs)"""
    kids = []
    while budget[0] > 0 and depth < 5 and rng.random() < (0.75 if depth < 2 else 0.4):
        budget[0] -= 1
        jt = str(rng.choice(ONE if rng.random() < 0.8 else ANY))
        child = (jt, rand_spec(depth + 1, budget))
        kids.append(child)
        if rng.random() < 0.5:  # the same shape once or twice more
            for _ in range(int(rng.integers(1, 3))):
                size = count(child)
                if budget[0] >= size:
                    budget[0] -= size
                    kids.append(child)
    return kids


def count(node):
    return 1 + sum(count(c) for c in node[1])


def sym(M):
    return np.tril(M) + np.transpose(np.tril(M, -1), (0, 2, 1))


worst, pairs_seen, done = {}, 0, 0
for trial in range(20 * N):
    if done >= N:
        break
    budget = [int(rng.integers(3, 16))]
    root = (str(rng.choice(["QuaternionFloating", "Revolute", "Fixed"])), rand_spec(1, budget))
    spec = [root] + ([(str(rng.choice(ONE)), rand_spec(1, budget))] if rng.random() < 0.3 else [])
    model = rbd.flatten(rbd.tree_mechanism(rng, spec, axis_aligned=bool(rng.random() < 0.5)))
    if model.nv == 0 or model.nv > 64 or model.n_bodies > 40:
        continue
    B = int(rng.integers(1, 80))
    r2 = np.random.default_rng(trial)
    q, v = rbd.rand_configuration(model, B, r2), rbd.rand_velocity(model, B, r2)
    tau, fe, vd = r2.random((B, model.nv)), r2.random((B, 6 * model.n_bodies)), r2.standard_normal((B, model.nv))

    def chk(name, got, ref, tol):
        e = float(np.abs(got - ref).max() / max(1.0, np.abs(ref).max()))
        worst[name] = max(worst.get(name, 0.0), e)
        assert np.isfinite(got).all() and e < tol, (trial, name, spec, B, e)

    try:
        src = rbd.jit_source(model, torch.float32, "dynamics")
    except rbd._capi.RBDError:  # (more than eight children on one body: outside the library's limits)
        continue
    if src is None:
        continue
    done += 1
    npair = int(re.search(r"NPAIR = (\d+)", src).group(1))
    pairs_seen += npair
    lib = spec_emu.build(src, "ABA")
    Ms, c = sym(oracle.mass_matrix(model, q)), None
    for f in (fe, None):
        vdg, qdg = spec_emu.aba_f32(lib, model, q, v, tau, f, want_qdot=True)
        cb = oracle.dynamics_bias(model, q, v, f)
        res = np.einsum("bij,bj->bi", Ms, vdg.astype(np.float64)) - (tau - cb)
        # backward error of M v̇ + c = τ in ALL its data (τ and c apart: where they nearly cancel, the rounding of c alone is large against τ − c)
        eta = float((np.linalg.norm(res, axis=1) / (np.linalg.norm(Ms, axis=(1, 2)) * np.linalg.norm(vdg, axis=1) + np.linalg.norm(tau, axis=1) + np.linalg.norm(cb, axis=1))).max())
        worst["dynamics! backward error f32"] = max(worst.get("dynamics! backward error f32", 0.0), eta)
        assert np.isfinite(vdg).all() and eta < 2e-5, (trial, spec, B, eta)  # (fp32: the terms of c are an order of magnitude above c itself)
        chk("q̇ f32", qdg, oracle.dynamics(model, q, v, tau, f, want_qdot=True)[1], 2e-6)
    for dt, which, npt, tol in (("f32", "RNEA_F32", np.float32, 3e-5), ("f64", "RNEA_F64", np.float64, 1e-10)):
        s2 = rbd.jit_source(model, torch.float32 if dt == "f32" else torch.float64, "inverse_dynamics")
        if s2 is None:
            continue
        l2 = spec_emu.build(s2, which)
        t_, acc, jw = spec_emu.rnea(l2, model, q, v, vd, fe, dtype=npt, want_bodies=True)
        tr, jwr, accr = oracle.inverse_dynamics_bodies(model, q, v, vd, fe)
        chk("inverse_dynamics " + dt, t_, tr, tol); chk("accelerations " + dt, acc, accr.reshape(B, -1), tol); chk("jointwrenches " + dt, jw, jwr.reshape(B, -1), tol)
        chk("dynamics_bias " + dt, spec_emu.rnea(l2, model, q, v, None, None, dtype=npt), oracle.dynamics_bias(model, q, v, None), tol)
    Mr = oracle.mass_matrix(model, q)
    il = np.tril_indices(model.nv)
    for dt, which, npt, tol in (("f32", "MASS_F32", np.float32, 2e-6), ("f64", "MASS_F64", np.float64, 1e-10)):
        s3 = rbd.jit_source(model, torch.float32 if dt == "f32" else torch.float64, "mass_matrix")
        l3 = spec_emu.build(s3, which)
        chk("mass_matrix " + dt, spec_emu.crba(l3, model, q, npt)[:, il[0], il[1]], Mr[:, il[0], il[1]], tol)
        if dt == "f32" and "RBD_SPEC_CHOL" in s3:
            perm = np.array([int(x) for x in re.search(r"constexpr int PERM\[NV\] = \{([^}]*)\}", s3).group(1).split(",")])
            pr, pc = perm[il[0]], perm[il[1]]
            chk("staging triangle f32", spec_emu.crba(l3, model, q, npt, permuted=True)[:, np.maximum(pr, pc), np.minimum(pr, pc)], Mr[:, il[0], il[1]], tol)
    print(done, f"bodies {model.n_bodies} nv {model.nv} pairs {npair} B {B}", flush=True)
print(f"{done} random trees through the emulated compiled kernels ok ({pairs_seen} bodies walked as the partner of another); worst", worst)
