"""Stress of the one-lane-per-state kernels compiled for the mechanism (csrc/rbd_spec.hpp) on random trees of EVERY tree joint type — Planar,
QuaternionSpherical and QuaternionFloating joints below other bodies included: inverse_dynamics! / dynamics_bias! (fp64, fp32, per-body outputs),
dynamics! (fp32, RBD_ALGO_ABA_COMPILED, q̇ included), mass_matrix! (crba_spec) and the Cholesky solve behind it, both layouts, ragged batches, against the oracle.
usage: python scripts/stress_compiled_alltypes.py [N=40] [--precompile]   (--precompile: no GPU — compile the N mechanisms' programs into the cache and exit)"""
import os, sys
os.environ.setdefault("RBD_JIT_ASYNC", "0")
os.environ["RBD_TUNE"] = "state_min_batch=1,spec_aba_min_batch=1"  # every batch size through the one-lane-per-state routes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import rbd_amd as rbd
args = [a for a in sys.argv[1:] if not a.startswith("--")]
N = int(args[0]) if args else 40
PRE = "--precompile" in sys.argv
if not PRE:
    import oracle, simulate_np
rng = np.random.default_rng(4242)
TYPES = ["Revolute", "Prismatic", "Fixed", "SinCosRevolute", "Planar", "QuaternionSpherical", "QuaternionFloating"]
worst, used = {}, {}
sym = lambda M: np.tril(M) + np.transpose(np.tril(M, -1), (0, 2, 1))
done = 0
for trial in range(10 * N):
    if done >= N:
        break
    n = int(rng.integers(2, 14))
    types = [str(rng.choice(TYPES, p=[0.25, 0.1, 0.1, 0.05, 0.2, 0.15, 0.15])) for _ in range(n)]
    cb = float(rng.uniform(0, 1))
    sel = lambda mech, r: mech.bodies[-1] if r.random() < cb else mech.bodies[r.integers(len(mech.bodies))]
    model = rbd.flatten(rbd.rand_tree_mechanism(rng, types, sel))
    if model.nv == 0 or model.nv > 64:
        continue
    done += 1
    if PRE:
        part = os.environ.get("STRESS_PART")  # "k/n": this process compiles every n-th mechanism (hiprtc compiles one program at a time per process)
        if part and (done - 1) % int(part.split("/")[1]) != int(part.split("/")[0]):
            continue
        for dt in (torch.float32, torch.float64):
            ok, log = rbd.jit_precompile(model, dt)
            assert ok, (types, log[-2000:])
        print(done, types, flush=True)
        continue
    layout = "aos" if trial % 2 else "soa"
    B = int(rng.integers(1, 150))
    r2 = np.random.default_rng(trial)
    q, v = rbd.rand_configuration(model, B, r2), rbd.rand_velocity(model, B, r2)
    tau, fe, vd = r2.random((B, model.nv)), r2.random((B, 6 * model.n_bodies)), r2.standard_normal((B, model.nv))
    nb, nv = model.n_bodies, model.nv

    def chk(name, got, ref, tol):
        e = np.abs(got - ref).max() / max(1.0, np.abs(ref).max())
        worst[name] = max(worst.get(name, 0.0), e)
        assert e < tol, (trial, name, types, B, layout, e)

    def kernel(state, what):
        k = rbd.last_kernel(state)
        assert what in k, (trial, types, what, k)
        used[what] = used.get(what, 0) + 1

    Mr = oracle.mass_matrix(model, q)
    Ms = sym(Mr)
    for dname, tdt, tol in (("f64", torch.float64, 1e-10), ("f32", torch.float32, 3e-5)):
        qd_, vd_, tau_, fe_ = (a.astype(np.float32).astype(np.float64) if dname == "f32" else a for a in (q, vd, tau, fe))
        v_ = v.astype(np.float32).astype(np.float64) if dname == "f32" else v
        state = rbd.MechanismState(model, B, dtype=tdt, layout=layout)
        rbd.set_configuration_(state, qd_); rbd.set_velocity_(state, v_)
        D = lambda a: (torch.as_tensor(a, dtype=tdt).cuda() if layout == "aos" else torch.as_tensor(np.ascontiguousarray(a.T), dtype=tdt).cuda())
        Hh = lambda t: (t.double().cpu().numpy() if layout == "aos" else t.double().cpu().numpy().T)
        out = torch.full_like(D(tau_), float("nan"))
        jw = torch.full_like(D(fe_), float("nan")); acc = torch.full_like(jw, float("nan"))
        rbd.inverse_dynamics_(out, state, D(vd_), D(fe_), mapping="compiled", jointwrenchesout=jw, accelerations=acc)
        kernel(state, "rnea_spec_" + dname)
        t_ref, jw_ref, acc_ref = oracle.inverse_dynamics_bodies(model, qd_, v_, vd_, fe_)
        chk("inverse_dynamics " + dname, Hh(out), t_ref, tol)
        chk("jointwrenches " + dname, Hh(jw).reshape(B, nb, 6), jw_ref, tol)
        chk("accelerations " + dname, Hh(acc).reshape(B, nb, 6), acc_ref, tol)
        rbd.dynamics_bias_(out, state, mapping="compiled")
        chk("dynamics_bias " + dname, Hh(out), oracle.dynamics_bias(model, qd_, v_, None), tol)
        if layout == "soa":  # crba_spec's own stores
            res = rbd.DynamicsResult(model, B, dtype=tdt, layout=layout)
            res.massmatrix.fill_(float("nan"))
            rbd.mass_matrix_(res, state)
            kernel(state, "crba_spec_" + dname)
            Mg = Hh(res.massmatrix).reshape(B, nv, nv).transpose(0, 2, 1)
            il = np.tril_indices(nv)
            assert np.isfinite(Mg[:, il[0], il[1]]).all(), (trial, types)
            chk("mass_matrix " + dname, np.tril(Mg), np.tril(oracle.mass_matrix(model, qd_)), 1e-10 if dname == "f64" else 3e-6)
        x = torch.zeros_like(D(tau_))
        rbd.mass_matrix_solve_(x, state, D(tau_), None)  # (fp32 state-major, nv a multiple of 4: crba_spec_perm + chol_spec)
        assert rbd.sync(state) == 0
        xg = Hh(x)
        Msq = sym(oracle.mass_matrix(model, qd_))
        r = np.einsum("bij,bj->bi", Msq, xg) - tau_
        eta = (np.linalg.norm(r, axis=1) / (np.linalg.norm(Msq, axis=(1, 2)) * np.linalg.norm(xg, axis=1) + np.linalg.norm(tau_, axis=1))).max()
        worst["solve backward error " + dname] = max(worst.get("solve backward error " + dname, 0.0), eta)
        assert eta < (1e-12 if dname == "f64" else 2e-5), (trial, types, eta, rbd.last_kernel(state))
        k = rbd.last_kernel(state).split(" (")[0]
        used[k] = used.get(k, 0) + 1
        # round 6: the kinematics by-products one lane per state (family 11), both scalar types
        os.environ["RBD_TUNE"] = "state_min_batch=1,spec_aba_min_batch=1,spec_kin_min_batch=1"
        stk = rbd.MechanismState(model, B, dtype=tdt, layout=layout)
        rbd.set_configuration_(stk, qd_); rbd.set_velocity_(stk, v_)
        A = torch.zeros((B, 6 * nv) if layout == "aos" else (6 * nv, B), dtype=tdt, device="cuda")
        rbd.momentum_matrix_(A, stk)
        if "kin_spec" in rbd.last_kernel(stk):
            used["kin_spec_" + dname] = used.get("kin_spec_" + dname, 0) + 1
            tk = 1e-11 if dname == "f64" else 3e-4
            A_ref, _, com_ref = oracle.momentum_matrix(model, qd_, v_)
            chk("momentum matrix " + dname, Hh(A).reshape(B, nv, 6).transpose(0, 2, 1), A_ref, tk)
            ke_ref, pe_ref = oracle.energy(model, qd_, v_)
            chk("kinetic energy " + dname, rbd.kinetic_energy(stk).double().cpu().numpy(), ke_ref, tk)
            chk("potential energy " + dname, rbd.gravitational_potential_energy(stk).double().cpu().numpy(), pe_ref, tk)
            chk("centre of mass " + dname, Hh(rbd.center_of_mass(stk)), com_ref, tk)
            h_ref, hb_ref = oracle.momentum(model, qd_, v_)
            chk("momentum " + dname, rbd.momentum(stk).double().cpu().numpy(), h_ref, 10 * tk)
            chk("momentum rate bias " + dname, rbd.momentum_rate_bias(stk).double().cpu().numpy(), hb_ref, 100 * tk)
            base, body = (int(x) for x in r2.choice(np.arange(-1, nb), 2, replace=False)) if nb > 1 else (-1, 0)
            J = torch.full_like(A, 7.0)
            rbd.geometric_jacobian_(J, stk, base, body)
            J_ref, _ = oracle.geometric_jacobian(model, qd_, base, body)
            chk("geometric jacobian " + dname, Hh(J).reshape(B, nv, 6).transpose(0, 2, 1), J_ref, tk)
        if dname == "f64" and rbd.jit_source(model, torch.float64, "dynamics") is not None:  # round 6: dynamics! in doubles for what no walk kernel takes
            ref64, qd_ref = oracle.dynamics(model, qd_, v_, tau_, fe_, want_qdot=True)
            tune0 = os.environ.get("RBD_TUNE")
            for stash in (0, 1):  # both programs: every row in LDS, and the spare rows in the workspace's HBM stash (csrc/rbd_spec.hpp aba_spec GST)
                os.environ["RBD_TUNE"] = (tune0 + "," if tune0 else "") + "spec_f64_stash=%d" % stash
                name = "aba_spec_gst_f64" if stash else "aba_spec_f64"
                st1 = rbd.MechanismState(model, B, dtype=tdt, layout=layout)
                rbd.set_configuration_(st1, qd_); rbd.set_velocity_(st1, v_)
                res = rbd.DynamicsResult(model, B, dtype=tdt, layout=layout)
                rbd.dynamics_(res, st1, D(tau_), D(fe_), algorithm="aba_compiled")
                kernel(st1, name)
                chk("dynamics! f64 " + name, Hh(res.vd), ref64, 1e-10)
                chk("q̇ f64 " + name, Hh(res.qd), qd_ref, 1e-13)
                rbd.dynamics_(res, st1, D(tau_), None, algorithm="aba_compiled")
                chk("dynamics! f64 without wrenches " + name, Hh(res.vd), oracle.dynamics(model, qd_, v_, tau_), 1e-10)
                st2 = rbd.MechanismState(model, B, dtype=tdt, layout=layout)
                rbd.set_configuration_(st2, qd_); rbd.set_velocity_(st2, v_)
                rbd.simulate_(st2, 1.5e-3, dt=1e-3, torques=D(tau_))
                kernel(st2, name + " with the Munthe-Kaas stage folded in")
                sel = [0, B - 1] if B > 1 else [0]
                _, q_ref, v_ref = simulate_np.simulate(model, qd_[sel], v_[sel], 1.5e-3, 1e-3, tau_[sel])
                chk("simulate q f64 (|.|: quaternion sign) " + name, np.abs(Hh(st2.q)[sel]), np.abs(q_ref), 1e-10)
                chk("simulate v f64 " + name, Hh(st2.v)[sel], v_ref, 1e-8)
            if tune0 is None: os.environ.pop("RBD_TUNE")
            else: os.environ["RBD_TUNE"] = tune0
        if dname == "f32":
            res = rbd.DynamicsResult(model, B, dtype=tdt, layout=layout)
            try:
                rbd.dynamics_(res, state, D(tau_), D(fe_), algorithm="aba_compiled")
            except rbd._capi.RBDError as e:
                assert e.status == 3, e  # (more rows than the static LDS limit holds)
                used["aba_spec_f32 not available"] = used.get("aba_spec_f32 not available", 0) + 1
                continue
            kernel(state, "aba_spec_f32")
            vg = Hh(res.vd)
            c = oracle.dynamics_bias(model, qd_, v_, fe_)
            r = np.einsum("bij,bj->bi", Msq, vg) - (tau_ - c)
            eta = (np.linalg.norm(r, axis=1) / (np.linalg.norm(Msq, axis=(1, 2)) * np.linalg.norm(vg, axis=1) + np.linalg.norm(tau_ - c, axis=1))).max()
            worst["dynamics! backward error f32"] = max(worst.get("dynamics! backward error f32", 0.0), eta)
            assert np.isfinite(vg).all() and eta < 2e-5, (trial, types, B, layout, eta)
            _, qd_ref = oracle.dynamics(model, qd_, v_, tau_, fe_, want_qdot=True)
            chk("q̇ f32", Hh(res.qd), qd_ref, 3e-6)
            # simulate: two RK4 steps with the Munthe-Kaas stage inside aba_spec_f32 (generic over the joint types), two states against the numpy integrator
            st2 = rbd.MechanismState(model, B, dtype=tdt, layout=layout)
            rbd.set_configuration_(st2, qd_); rbd.set_velocity_(st2, v_)
            rbd.simulate_(st2, 1.5e-3, dt=1e-3, torques=D(tau_))
            kernel(st2, "aba_spec_f32 with the Munthe-Kaas stage folded in")
            qs_, vs_ = Hh(st2.q), Hh(st2.v)
            assert np.isfinite(qs_).all() and np.isfinite(vs_).all(), (trial, types)
            sel = [0, B - 1] if B > 1 else [0]
            _, q_ref, v_ref = simulate_np.simulate(model, qd_[sel], v_[sel], 1.5e-3, 1e-3, tau_[sel])
            chk("simulate q f32 (|.|: quaternion sign)", np.abs(qs_[sel]), np.abs(q_ref), 2e-5)
            chk("simulate v f32", vs_[sel], v_ref, 2e-3)
print(f"{done} random trees of all joint types through the compiled kernels ok; worst {worst}; kernels {used}")
