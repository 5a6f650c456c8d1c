"""One-off stress of the walk kernel compiled per mechanism at run time (aba_walk_spec and rnea_walk_spec, csrc/rbd_walk.hpp, fp64): random tree topologies — chains, bushes,
fixed / prismatic / sin-cos joints, with and without a 6-dof root (re-rooted plans) — plus the golden humanoids, against the oracle at 1e-10, and against
the interpreting walk kernel bit for bit.
    python scripts/stress_walk_compiled.py N --precompile     (no GPU: compiles the N trees' programs into the library's cache; ~1 min per big tree)
    python scripts/stress_walk_compiled.py N                  (GPU)"""
import os, sys
os.environ.setdefault("RBD_JIT_ASYNC", "0")  # wait for the kernels compiled per mechanism instead of starting on the interpreting ones
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
os.environ["RBD_TUNE"] = "spec_walk_min_batch=1"
import numpy as np, torch
import rbd_amd as rbd
from test_chain_plan import random_tree
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
pre = "--precompile" in sys.argv
rng = np.random.default_rng(99)
cases = [(name, rbd.load_flat_model(os.path.join(ROOT, "tests", "golden", "models", name + ".json"))) for name in ("atlas_fixed", "valkyrie_floating")]
for trial in range(N):
    n = int(rng.integers(1, 30))
    cases.append(("tree%d(%d bodies%s)" % (trial, n, ", floating" if trial % 2 else ""), rbd.flatten(random_tree(rbd, rng, n, bool(trial % 2), float(rng.uniform(0, 1))))))
if pre:
    import time
    for name, model in cases:
        t = time.time()
        src = rbd.jit_source(model, torch.float64, "dynamics_tracks")
        ok = (rbd.jit_precompile(model, torch.float64)[0], rbd.jit_precompile(model, torch.float32)[0]) if src else None
        print(name, "nv", model.nv, "program" if src else "outside the walk mapping", ok, round(time.time() - t, 1), "s", flush=True)
    sys.exit(0)
import oracle
done, worst, skipped = 0, 0.0, 0
for trial, (name, model) in enumerate(cases):
    if model.nv == 0 or rbd.jit_source(model, torch.float64, "dynamics_tracks") is None:
        skipped += 1
        continue
    B = int(np.random.default_rng(trial).integers(1, 200))
    r2 = np.random.default_rng(1000 + trial)
    q, v = rbd.rand_configuration(model, B, r2), rbd.rand_velocity(model, B, r2)
    tau, fe = r2.random((B, model.nv)), r2.random((B, 6 * model.n_bodies))
    t = lambda a: torch.as_tensor(a, dtype=torch.float64, device="cuda")
    outs = {}
    for jit in ("1", "0"):
        os.environ["RBD_JIT"] = jit
        state = rbd.MechanismState(model, B); res = rbd.DynamicsResult(model, B)
        rbd.set_configuration_(state, q); rbd.set_velocity_(state, v)
        try:
            rbd.dynamics_(res, state, t(tau), t(fe), algorithm="aba_walk")
        except rbd._capi.RBDError as ex:
            assert ex.status == 3
            break
        assert rbd.sync(state) == 0
        kern = rbd.last_kernel(state)
        back = torch.zeros_like(state.v); jw = torch.zeros(B, 6 * model.n_bodies, dtype=torch.float64, device="cuda"); acc = torch.zeros_like(jw)
        rbd.inverse_dynamics_(back, state, res.vd, t(fe), mapping="walk", jointwrenchesout=jw, accelerations=acc)
        assert rbd.sync(state) == 0 and ("rnea_walk_spec" if jit == "1" else "rnea_walk_kernel") in rbd.last_kernel(state), rbd.last_kernel(state)
        t_ref, jw_ref, acc_ref = oracle.inverse_dynamics_bodies(model, q, v, res.vd.cpu().numpy(), fe)
        for got, ref_ in ((back, t_ref), (jw, jw_ref.reshape(B, -1)), (acc, acc_ref.reshape(B, -1))):
            assert np.abs(got.cpu().numpy() - ref_).max() <= 1e-10 * max(1.0, np.abs(ref_).max()), (name, jit)
        outs[jit] = (res.vd.cpu().numpy().copy(), res.qd.cpu().numpy().copy(), kern)
    if len(outs) < 2:
        skipped += 1
        continue
    assert "aba_walk_spec" in outs["1"][2] and "aba_walk_kernel" in outs["0"][2], (name, outs["1"][2], outs["0"][2])
    ref, qd = oracle.dynamics(model, q, v, tau, fe, want_qdot=True)
    e = np.abs(outs["1"][0] - ref).max() / max(1.0, np.abs(ref).max())
    e0 = np.abs(outs["0"][0] - ref).max() / max(1.0, np.abs(ref).max())
    worst = max(worst, e)
    # same arithmetic in the same order; contraction of a * b + c may differ between the two compilations
    assert e < 1e-10 and np.abs(outs["1"][1] - qd).max() <= 1e-12 * max(1.0, np.abs(qd).max()), (name, e)
    # fp32, one state per lane and two: against the interpreting kernel of the same form (same arithmetic up to contraction) and the fp64 oracle at fp32 accuracy
    f32 = lambda a: a.astype(np.float32).astype(np.float64)
    q32, v32, tau32, fe32 = f32(q), f32(v), f32(tau), f32(fe)
    ref32 = oracle.dynamics(model, q32, v32, tau32, fe32)
    e32 = {}
    for pair in ("1000000000", "1"):
        os.environ["RBD_TUNE"] = "spec_walk_min_batch=1,walk_pair_min_batch=" + pair
        got = {}
        for jit in ("1", "0"):
            os.environ["RBD_JIT"] = jit
            state = rbd.MechanismState(model, B, dtype=torch.float32); res = rbd.DynamicsResult(model, B, dtype=torch.float32)
            rbd.set_configuration_(state, q32); rbd.set_velocity_(state, v32)
            t32 = lambda a: torch.as_tensor(a, dtype=torch.float32, device="cuda")
            rbd.dynamics_(res, state, t32(tau32), t32(fe32), algorithm="aba_walk")
            assert rbd.sync(state) == 0
            assert (("aba_walk_spec" if jit == "1" else "aba_walk_kernel") in rbd.last_kernel(state)) and (("two fp32" in rbd.last_kernel(state)) == (pair == "1")), rbd.last_kernel(state)
            got[jit] = res.vd.double().cpu().numpy()
        scale = max(1.0, np.abs(ref32).max())
        e32[pair == "1"] = (np.abs(got["1"] - ref32).max() / scale, np.abs(got["0"] - ref32).max() / scale)
        # an fp32 solve of an ill-conditioned system: the two compilations may differ by as much as either differs from the truth, not more
        assert e32[pair == "1"][0] <= 4 * e32[pair == "1"][1] + 1e-5, (name, pair, e32)
    os.environ["RBD_TUNE"] = "spec_walk_min_batch=1"
    print(name, "nv", model.nv, "B", B, "compiled err %.1e interpreting err %.1e | fp32 (compiled, interpreting): one per lane %.1e %.1e two per lane %.1e %.1e" % ((e, e0) + tuple(float(x) for x in e32[False] + e32[True])), flush=True)
    done += 1
print(done, "mechanisms ok,", skipped, "outside the mapping; worst error", worst)
