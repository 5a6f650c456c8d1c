"""One-off stress of the two-bodies-per-lane kernels compiled per mechanism (csrc/rbd_jit.hip spec_bank_source: aba_bank_spec / aba_bank_fused_spec / rnea_bank_spec with the
level loops unrolled against the mechanism's level structure): random tree topologies — chains, bushes, deep and shallow, with and without a 6-dof root, fixed /
prismatic / sin-cos joints (the generic instantiation) and all-revolute ones (SIMPLE) — against the oracle at 1e-10 and against the kernels built with the library.
    python scripts/stress_bank_compiled.py N --precompile [k/n]   (no GPU: compiles the trees' banked programs into the library's cache, every n-th tree)
    python scripts/stress_bank_compiled.py N                      (GPU)"""
import os, sys, time
os.environ.setdefault("RBD_JIT_ASYNC", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import rbd_amd as rbd
from test_chain_plan import random_tree
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
pre = "--precompile" in sys.argv
part, parts = (int(x) for x in sys.argv[sys.argv.index("--precompile") + 1].split("/")) if pre and len(sys.argv) > sys.argv.index("--precompile") + 1 else (0, 1)
rng = np.random.default_rng(404)
trees = []
for trial in range(N):
    n = int(rng.integers(2, 45))
    branch = float(rng.uniform(0, 1)) if trial % 3 else 0.0   # every third tree: a plain chain (as many levels as bodies)
    mech = random_tree(rbd, rng, n, bool(rng.integers(2)), branch)
    model = rbd.flatten(mech)
    if model.nv == 0 or rbd.bank_plan(model) is None or not rbd.bank_plan(model)["aba"]:
        continue
    trees.append((trial, model))
print(len(trees), "of", N, "trees in the banked scope; levels", [int(m.levels().max()) + 1 for _, m in trees])
if pre:
    t0 = time.time()
    for k, (trial, model) in enumerate(trees):
        if k % parts != part:
            continue
        for dt in (torch.float64, torch.float32):
            while (st := rbd.jit_status(model, dt, 8)) == 0:
                time.sleep(0.1)
            assert st == 1, (trial, dt)
    print("compiled in", round(time.time() - t0, 1), "s")
    sys.exit(0)
import oracle, simulate_np
worst = {}
for trial, model in trees:
    B = int(rng.integers(1, 150))
    r2 = np.random.default_rng(trial)
    q, v = rbd.rand_configuration(model, B, r2), rbd.rand_velocity(model, B, r2)
    tau, fe, vd = r2.random((B, model.nv)), r2.random((B, 6 * model.n_bodies)), r2.standard_normal((B, model.nv))
    ref, ref_id = oracle.dynamics(model, q, v, tau, fe), oracle.inverse_dynamics(model, q, v, vd, fe)
    got = {}
    for jit in ("1", "0"):
        os.environ["RBD_JIT"] = jit
        state = rbd.MechanismState(model, B); res = rbd.DynamicsResult(model, B)
        rbd.set_configuration_(state, q); rbd.set_velocity_(state, v)
        t, f = torch.as_tensor(tau).cuda(), torch.as_tensor(fe).cuda()
        rbd.dynamics_(res, state, t, f, algorithm="aba_banks")
        assert ("compiled" in rbd.last_kernel(state)) == (jit == "1"), (trial, jit, rbd.last_kernel(state))
        a = res.vd.cpu().numpy()
        err = np.abs(a - ref).max() / max(1.0, np.abs(ref).max())
        worst["aba " + jit] = max(worst.get("aba " + jit, 0.0), err)
        assert err < 1e-8, (trial, jit, err)
        out = torch.zeros_like(t)
        rbd.inverse_dynamics_(out, state, torch.as_tensor(vd).cuda(), f, mapping="banks")
        b = out.cpu().numpy()
        err = np.abs(b - ref_id).max() / max(1.0, np.abs(ref_id).max())
        worst["rnea " + jit] = max(worst.get("rnea " + jit, 0.0), err)
        assert err < 1e-9, (trial, jit, err)
        got[jit] = (a, b)
    assert np.abs(got["1"][0] - got["0"][0]).max() <= 1e-9 * max(1.0, np.abs(ref).max()) and np.abs(got["1"][1] - got["0"][1]).max() <= 1e-10 * max(1.0, np.abs(ref_id).max())
    if trial % 4 == 0:  # the integrator-fused instantiation: two RK4 steps against the numpy Munthe-Kaas restatement on three states
        os.environ["RBD_JIT"] = "1"
        os.environ["RBD_TUNE"] = "bank_min_batch=1"
        s2 = rbd.MechanismState(model, B)
        rbd.set_configuration_(s2, q); rbd.set_velocity_(s2, v)
        rbd.simulate_(s2, 1.5e-3, dt=1e-3, torques=torch.as_tensor(tau).cuda())
        assert rbd.last_kernel(s2).startswith("aba_bank_kernel") and "compiled" in rbd.last_kernel(s2), rbd.last_kernel(s2)
        n3 = min(B, 3)
        _, q_ref, v_ref = simulate_np.simulate(model, q[:n3], v[:n3], 1.5e-3, 1e-3, tau[:n3])
        e = max(np.abs(np.abs(s2.q.cpu().numpy()[:n3]) - np.abs(q_ref)).max(), np.abs(s2.v.cpu().numpy()[:n3] - v_ref).max() / max(1.0, np.abs(v_ref).max()))
        worst["simulate"] = max(worst.get("simulate", 0.0), e)
        assert e < 1e-8, (trial, e)
        os.environ.pop("RBD_TUNE")
    # fp32 on every fifth tree: backward error
    if trial % 5 == 0:
        os.environ["RBD_JIT"] = "1"
        s32 = rbd.MechanismState(model, B, dtype=torch.float32); r32 = rbd.DynamicsResult(model, B, dtype=torch.float32)
        rbd.set_configuration_(s32, q); rbd.set_velocity_(s32, v)
        rbd.dynamics_(r32, s32, torch.as_tensor(tau, dtype=torch.float32).cuda(), torch.as_tensor(fe, dtype=torch.float32).cuda(), algorithm="aba_banks")
        assert "compiled" in rbd.last_kernel(s32)
        back = oracle.inverse_dynamics(model, q, v, r32.vd.double().cpu().numpy(), fe)
        cb = oracle.dynamics_bias(model, q, v, fe)
        berr = (np.linalg.norm(back - tau, axis=1) / np.linalg.norm(tau - cb, axis=1)).max()
        worst["aba f32 backward"] = max(worst.get("aba f32 backward", 0.0), berr)
        assert berr < 2e-4, (trial, berr)
print(f"{len(trees)} random trees ok; worst relative errors {worst}")
