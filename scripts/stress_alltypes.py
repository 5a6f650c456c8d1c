"""One-off stress: random trees with EVERY tree joint type (incl. Planar, QuaternionSpherical, 6-dof joints below other bodies),
both layouts, fp64 — dynamics!, inverse_dynamics! (lanes / banks), mass_matrix!, kinematics by-products against the oracle.
--compiled: inverse_dynamics! through the kernel compiled for the mechanism as well (RBD_ALGO_ABA_COMPILED; one hiprtc compilation per tree — the fp32 / dynamics! /
mass-matrix side of the compiled kernels on such trees is scripts/stress_compiled_alltypes.py)."""
import os, sys
os.environ.setdefault("RBD_JIT_ASYNC", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import rbd_amd as rbd, oracle
COMPILED = "--compiled" in sys.argv
_a = [a for a in sys.argv[1:] if not a.startswith("--")]
N = int(_a[0]) if _a else 200
rng = np.random.default_rng(777)
TYPES = ["Revolute", "Prismatic", "Fixed", "SinCosRevolute", "Planar", "QuaternionSpherical", "QuaternionFloating"]
worst, skipped = {}, 0
for trial in range(N):
    n = int(rng.integers(1, 14))
    types = [str(rng.choice(TYPES, p=[0.3, 0.15, 0.1, 0.1, 0.15, 0.1, 0.1])) for _ in range(n)]
    cb = float(rng.uniform(0, 1))
    sel = lambda mech, r: mech.bodies[-1] if r.random() < cb else mech.bodies[r.integers(len(mech.bodies))]
    model = rbd.flatten(rbd.rand_tree_mechanism(rng, types, sel))
    if model.nv == 0 or model.nv > 64:
        continue
    layout = "aos" if trial % 2 else "soa"
    B = int(rng.integers(1, 40))
    r2 = np.random.default_rng(trial)
    q, v = rbd.rand_configuration(model, B, r2), rbd.rand_velocity(model, B, r2)
    tau, fe, vd = r2.random((B, model.nv)), r2.random((B, 6 * model.n_bodies)), r2.standard_normal((B, model.nv))
    try:
        state = rbd.MechanismState(model, B, layout=layout)
    except Exception:
        skipped += 1
        continue
    res = rbd.DynamicsResult(model, B, layout=layout)
    rbd.set_configuration_(state, q); rbd.set_velocity_(state, v)
    D = lambda a: (torch.as_tensor(a).cuda() if layout == "aos" else torch.as_tensor(np.ascontiguousarray(a.T)).cuda())
    Hh = lambda t: (t.cpu().numpy() if layout == "aos" else t.cpu().numpy().T)
    def chk(name, got, ref, tol):
        e = np.abs(got - ref).max() / max(1.0, np.abs(ref).max())
        worst[name] = max(worst.get(name, 0.0), e)
        assert e < tol, (trial, name, types, B, layout, e)
    rbd.dynamics_(res, state, D(tau), D(fe))
    chk("dynamics", Hh(res.vd), oracle.dynamics(model, q, v, tau, fe), 1e-8)
    out = torch.zeros_like(D(tau))
    ref = oracle.inverse_dynamics(model, q, v, vd, fe)
    for mp in ("lanes", "banks") + (("compiled",) if COMPILED else ()):
        try:
            rbd.inverse_dynamics_(out, state, D(vd), D(fe), mapping=mp)
        except Exception:
            continue
        chk("rnea_" + mp, Hh(out), ref, 1e-10)
    rbd.mass_matrix_(res, state)
    Mg = Hh(res.massmatrix).reshape(B, model.nv, model.nv).transpose(0, 2, 1)
    chk("mass_matrix", np.tril(Mg), np.tril(oracle.mass_matrix(model, q)), 1e-10)
    A_ref, _, com_ref = oracle.momentum_matrix(model, q, v)
    A = torch.zeros_like(D(np.zeros((B, 6 * model.nv))))
    rbd.momentum_matrix_(A, state)
    chk("momentum_matrix", Hh(A).reshape(B, model.nv, 6).transpose(0, 2, 1), A_ref, 1e-10)
    chk("com", Hh(rbd.center_of_mass(state)), com_ref, 1e-11)
    chk("momentum_rate_bias", rbd.momentum_rate_bias(state).cpu().numpy(), oracle.momentum(model, q, v)[1], 1e-9)
print(f"{N} random trees of all joint types ok; worst relative errors {worst}; outside the limits: {skipped}")
