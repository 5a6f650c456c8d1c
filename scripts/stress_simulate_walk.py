"""One-off stress of `simulate` through the walk kernels compiled per mechanism — the kernel that takes the four stages of a step in one launch
(aba_walk_sim_spec, csrc/rbd_walk.hpp; admitted statically since round 6: csrc/rbd_jit.hip jit_walk_admit) and the four-launch
route beside it: random trees of 1-dof joints (revolute, prismatic, sin-cos, fixed; chains and bushes) with and without a 6-dof root, fp64 and fp32, both
layouts, constant torques with and without external wrenches, against the numpy restatement of the integrator (oracle/simulate_np.py).
    python scripts/stress_simulate_walk.py N --precompile     (no GPU: the N trees' programs into the library's cache)
    python scripts/stress_simulate_walk.py N                  (GPU)"""
import os, sys
os.environ.setdefault("RBD_JIT_ASYNC", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import rbd_amd as rbd
from test_chain_plan import random_tree
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
pre = "--precompile" in sys.argv
rng = np.random.default_rng(515)
cases = []
for trial in range(N):
    n = int(rng.integers(2, 26))
    cases.append(("tree%d(%d bodies%s)" % (trial, n, ", floating" if trial % 2 else ""), rbd.flatten(random_tree(rbd, rng, n, bool(trial % 2), float(rng.uniform(0, 1))))))
if pre:
    import time
    for name, model in cases:
        t = time.time()
        src = rbd.jit_source(model, torch.float64, "dynamics_tracks_sim")
        ok = (rbd.jit_precompile(model, torch.float64)[0], rbd.jit_precompile(model, torch.float32)[0]) if src else None
        print(name, "nv", model.nv, "program" if src else "outside the walk mapping", ok, round(time.time() - t, 1), "s", flush=True)
    sys.exit(0)
import simulate_np
from test_gpu_parity import canon_q, host, dev
one, four, other, worst = 0, 0, 0, {"f64": 0.0, "f32": 0.0}
knobs = lambda one_launch, pairs: "walk_min_batch=1,spec_walk_min_batch=1,spec_aba_min_batch=1099511627776,sim_one_launch=%d,walk_pair_min_batch=%s" % (one_launch, "1" if pairs else "1099511627776")
for trial, (name, model) in enumerate(cases):
    if model.nv == 0 or rbd.jit_source(model, torch.float64, "dynamics_tracks_sim") is None:
        continue
    for dtype, layout, wrenches in (("f64", "aos", False), ("f64", "soa", True), ("f32", "aos", True), ("f32", "soa", False)):
        pairs = dtype == "f32" and trial % 3 == 0
        tdt = torch.float64 if dtype == "f64" else torch.float32
        B, dt, nsteps = int(np.random.default_rng(trial).integers(65, 260)), 1e-3, 2
        r2 = np.random.default_rng(2000 + trial)
        q, v, tau = rbd.rand_configuration(model, B, r2), 0.5 * rbd.rand_velocity(model, B, r2), r2.random((B, model.nv))
        fe = r2.random((B, 6 * model.n_bodies)) if wrenches else None
        got = {}
        for one_launch in (1, 0):
            os.environ["RBD_TUNE"] = knobs(one_launch, pairs)  # (read when the workspace is created)
            state = rbd.MechanismState(model, B, dtype=tdt, layout=layout)
            rbd.set_configuration_(state, q); rbd.set_velocity_(state, v)
            rbd.simulate_(state, (nsteps - 0.5) * dt, dt=dt, torques=dev(tau, state), externalwrenches=dev(fe, state) if wrenches else None)
            k = rbd.last_kernel(state)
            if "four stages per launch" in k: one += 1
            elif "folded in" in k: four += 1
            else: other += 1
            assert one_launch == 0 or "four stages per launch" in k or "folded in" not in k, (name, k)  # (a looped program the static admission turned down shows here)
            got[one_launch] = (host(state.q, state).astype(np.float64), host(state.v, state).astype(np.float64), k)
            assert np.isfinite(got[one_launch][0]).all() and np.isfinite(got[one_launch][1]).all(), (name, dtype, layout, k)
        # the two routes against each other (the same arithmetic from two compilations) ...
        tol2 = 1e-11 if dtype == "f64" else 1e-4
        for a, b in zip(got[1][:2], got[0][:2]):
            assert np.abs(a - b).max() <= tol2 * max(1.0, np.abs(b).max()), (name, dtype, layout, wrenches, got[1][2], got[0][2], np.abs(a - b).max())
        # ... and, without external wrenches (the restatement takes none), against the numpy integrator
        if not wrenches:
            sel = np.r_[0:2, B - 2:B]
            _, q_ref, v_ref = simulate_np.simulate(model, q[sel], v[sel], (nsteps - 0.5) * dt, dt, tau[sel])
            eq = np.abs(canon_q(model, got[1][0][sel]) - canon_q(model, q_ref)).max() / max(1.0, np.abs(q_ref).max())
            ev = np.abs(got[1][1][sel] - v_ref).max() / max(1.0, np.abs(v_ref).max())
            worst[dtype] = max(worst[dtype], eq, ev)
            assert max(eq, ev) <= (1e-9 if dtype == "f64" else 5e-3), (name, dtype, layout, got[1][2], eq, ev)
print("stress_simulate_walk: %d trees; calls through: four stages per launch %d, four launches %d, other kernels %d; worst rel err vs the numpy integrator fp64 %.2e fp32 %.2e" % (len(cases), one, four, other, worst["f64"], worst["f32"]))
