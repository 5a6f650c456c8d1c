"""One-lane-per-state kernels against the oracle: the interpreting ones (csrc/rbd_state.hpp: crba_state_kernel, rnea_state_kernel) and the
ones compiled for the mechanism at run time (csrc/rbd_spec.hpp through rbd_jit.hip: crba_spec, chol_spec, emit_spec; RBD_JIT=0 switches them
off).  They serve large batches (default: from half a chip-full of wavefronts up); RBD_TUNE state_min_batch=1 routes every batch size through them
here, so that the same small seeded cases the lane-per-body kernels are tested on apply.
reference: src/mechanism_algorithms.jl:248-272, :387-459, :542-553, :764, :819."""
import os

import numpy as np
import pytest
import torch

from conftest import LIMBS, tune
from test_gpu_parity import NT, TD, dev, host, make

pytestmark = pytest.mark.gpu
IN_SCOPE = ["atlas_floating", "atlas_fixed", "valkyrie_floating", "double_pendulum", "acrobot_urdf", "quickstart_pendulum"]
# Planar / QuaternionSpherical joints, QuaternionFloating joints below the world (the reference's randmech, test/test_mechanism_algorithms.jl:1-11): the
# kernels compiled for the mechanism take them, the interpreting one-lane-per-state kernels do not (the lane-per-body kernels stand behind)
EVERY_JOINT_TYPE = ["randmech1", "randmech2", "randmech3", "inner_floating", "mixed20"]


@pytest.fixture(params=["compiled", "interpreted"])
def states_everywhere(monkeypatch, request):
    tune(monkeypatch, state_min_batch="1")
    monkeypatch.setenv("RBD_JIT", "1" if request.param == "compiled" else "0")  # read when a workspace first needs the kernels
    return request.param


def sym(M):
    return np.tril(M) + np.transpose(np.tril(M, -1), (0, 2, 1))


@pytest.mark.parametrize("layout", ["aos", "soa"])
@pytest.mark.parametrize("name", IN_SCOPE + EVERY_JOINT_TYPE)
def test_state_kernels_f64(rbd, oracle, models, name, layout, states_everywhere):
    model = models[name]
    B, nv = 70, model.nv  # two wavefronts, the second partly filled
    state, q, v, tau, fe = make(rbd, model, B, "f64", layout, 41)
    rng = np.random.default_rng(3)
    vd = rng.standard_normal((B, nv))
    # inverse_dynamics! with v̇ and external wrenches; dynamics_bias!
    out = torch.zeros_like(state.v)
    rbd.inverse_dynamics_(out, state, dev(vd, state), dev(fe, state))
    ref = oracle.inverse_dynamics(model, q, v, vd, fe)
    assert np.abs(host(out, state) - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max())
    rbd.dynamics_bias_(out, state)
    ref = oracle.dynamics_bias(model, q, v, None)
    assert np.abs(host(out, state) - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max())
    # mass_matrix! (batch-innermost layout: straight from the state kernel), structural zeros included
    result = rbd.DynamicsResult(model, B, layout=layout)
    result.massmatrix.fill_(float("nan"))
    rbd.mass_matrix_(result, state)
    got = host(result.massmatrix, state).reshape(B, nv, nv).transpose(0, 2, 1)
    Mr = oracle.mass_matrix(model, q)
    il = np.tril_indices(nv)
    assert np.isfinite(got[:, il[0], il[1]]).all()
    assert np.abs(got[:, il[0], il[1]] - Mr[:, il[0], il[1]]).max() <= 1e-10 * max(1.0, np.abs(Mr).max())
    # dynamics! through the reference's route: bias + mass matrix + Cholesky, q̇ included
    rbd.dynamics_(result, state, dev(tau, state), dev(fe, state), algorithm="crba")
    ref, qd_ref = oracle.dynamics(model, q, v, tau, fe, want_qdot=True)
    assert np.abs(host(result.vd, state) - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max())
    assert np.abs(host(result.qd, state) - qd_ref).max() <= 1e-13 * max(1.0, np.abs(qd_ref).max())
    assert rbd.sync(state) == 0


@pytest.mark.parametrize("layout", ["aos", "soa"])
@pytest.mark.parametrize("name", ["atlas_floating", "valkyrie_floating", "double_pendulum", "mixed20", "randmech1"] + LIMBS)
def test_state_kernels_f32_solve(rbd, oracle, models, name, layout, states_everywhere):
    """fp32: mass_matrix! + Cholesky solve (BASELINE configs[2] shape).  AOS callers get M re-emitted by the tile Cholesky from the staging copy."""
    model = models[name]
    B, nv = 200, model.nv
    state, q, v, tau, fe = make(rbd, model, B, "f32", layout, 43)
    x = torch.zeros_like(state.v)
    Mout = torch.full(((B, nv * nv) if layout == "aos" else (nv * nv, B)), float("nan"), dtype=torch.float32, device="cuda")
    rbd.mass_matrix_solve_(x, state, dev(tau, state), Mout)
    assert rbd.sync(state) == 0
    Mr = oracle.mass_matrix(model, q)
    got = host(Mout, state).reshape(B, nv, nv).transpose(0, 2, 1)
    il = np.tril_indices(nv)
    assert np.isfinite(got[:, il[0], il[1]]).all()
    assert np.abs(got[:, il[0], il[1]] - Mr[:, il[0], il[1]]).max() <= 2e-6 * np.abs(Mr).max()
    Ms, xg = sym(Mr), host(x, state)
    res = np.einsum("bij,bj->bi", Ms, xg) - tau
    eta = np.linalg.norm(res, axis=1) / (np.linalg.norm(Ms, axis=(1, 2)) * np.linalg.norm(xg, axis=1) + np.linalg.norm(tau, axis=1))
    assert eta.max() <= 1e-5, eta.max()
    # dynamics! (reference route) in fp32: backward error of M v̇ = τ − c
    result = rbd.DynamicsResult(model, B, dtype=torch.float32, layout=layout)
    rbd.dynamics_(result, state, dev(tau, state), dev(fe, state), algorithm="crba")
    cr = oracle.dynamics_bias(model, q, v, fe)
    vg = host(result.vd, state)
    res = np.einsum("bij,bj->bi", Ms, vg) - (tau - cr)
    eta = np.linalg.norm(res, axis=1) / (np.linalg.norm(Ms, axis=(1, 2)) * np.linalg.norm(vg, axis=1) + np.linalg.norm(tau - cr, axis=1))
    assert eta.max() <= 2e-5, eta.max()


@pytest.mark.gpu
def test_mass_matrix_solve_f32_takes_the_compiled_kernels_from_small_batches(rbd, oracle, models):
    """Round 6: fp32 `mass_matrix!` + Cholesky solve (BASELINE configs[2]'s call) runs on the two kernels compiled for the mechanism from 256 states on — measured
    ahead of the one-body-per-lane kernels at every batch (scripts/exp_mass_small.sh: 256 states 28 against 40 us, 16 384: 46 against 117) — instead of from
    32 768; `mass_matrix!` alone from 10 241.  No RBD_TUNE here: the library's own thresholds."""
    model = models["atlas_floating"]
    nv = model.nv
    for B, compiled in ((200, False), (300, True)):
        state, q, v, tau, fe = make(rbd, model, B, "f32", "aos", 47)
        x = torch.zeros_like(state.v)
        Mout = torch.full((B, nv * nv), float("nan"), dtype=torch.float32, device="cuda")
        try:
            rbd.mass_matrix_solve_(x, state, dev(tau, state), Mout)
        except rbd._capi.RBDError as e:
            if e.status == 3:
                pytest.skip("hiprtc not available")
            raise
        k = rbd.last_kernel(state)
        if compiled and "chol_spec" not in k and not rbd.jit_precompile(models["double_pendulum"], torch.float32)[0]:
            pytest.skip("hiprtc not available")
        assert rbd.sync(state) == 0 and ("chol_spec_f32" in k) == compiled, (B, k)
        Mr = oracle.mass_matrix(model, q)
        got = host(Mout, state).reshape(B, nv, nv).transpose(0, 2, 1)
        il = np.tril_indices(nv)
        assert np.abs(got[:, il[0], il[1]] - Mr[:, il[0], il[1]]).max() <= 2e-6 * np.abs(Mr).max()
        Ms, xg = sym(Mr), host(x, state)
        res = np.einsum("bij,bj->bi", Ms, xg) - tau
        eta = np.linalg.norm(res, axis=1) / (np.linalg.norm(Ms, axis=(1, 2)) * np.linalg.norm(xg, axis=1) + np.linalg.norm(tau, axis=1))
        assert eta.max() <= 1e-5, eta.max()


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["dense", "no_M", "packed"])
def test_first_use_check_of_the_compiled_mass_matrix_solve(rbd, oracle, models, form, monkeypatch, capfd):
    """The first x a workspace gets from the fp32 pair compiled for the mechanism (crba_spec_perm + chol_spec, its M_out = NULL and packed-triangle forms) is
    compared with crba_kernel + the dense Cholesky kernel on the call's first states (csrc/rbd_capi.hip first_use_check_solve); here every check is made to find
    a difference (RBD_TUNE first_use_inject=1): the pair is dropped, the call recomputed, x and M are right, a message says so."""
    model = models["atlas_floating"]
    nv = model.nv
    tune(monkeypatch, first_use_inject=1)
    B = 300
    state, q, v, tau, fe = make(rbd, model, B, "f32", "aos", 48)
    x = torch.zeros_like(state.v)
    Mr = oracle.mass_matrix(model, q)
    il = np.tril_indices(nv)
    if form == "packed":
        P = torch.full((B, nv * (nv + 1) // 2), float("nan"), dtype=torch.float32, device="cuda")
        rbd.mass_matrix_solve_(x, state, dev(tau, state), P, packed=True)
    else:
        Mout = torch.full((B, nv * nv), float("nan"), dtype=torch.float32, device="cuda") if form == "dense" else None
        rbd.mass_matrix_solve_(x, state, dev(tau, state), Mout)
    k = rbd.last_kernel(state)
    err = capfd.readouterr().err
    if "first_use_inject" not in err and not rbd.jit_precompile(models["double_pendulum"], torch.float32)[0]:
        pytest.skip("hiprtc not available")
    assert rbd.sync(state) == 0 and "chol_spec" not in k and "first_use_inject" in err, (k, err)
    Ms, xg = sym(Mr), host(x, state)
    res = np.einsum("bij,bj->bi", Ms, xg) - tau
    eta = np.linalg.norm(res, axis=1) / (np.linalg.norm(Ms, axis=(1, 2)) * np.linalg.norm(xg, axis=1) + np.linalg.norm(tau, axis=1))
    assert eta.max() <= 1e-5, eta.max()
    if form == "dense":
        got = host(Mout, state).reshape(B, nv, nv).transpose(0, 2, 1)
        assert np.abs(got[:, il[0], il[1]] - Mr[:, il[0], il[1]]).max() <= 2e-6 * np.abs(Mr).max()
    # the second call: the kernels built with the library, no check
    if form == "packed":
        rbd.mass_matrix_solve_(x, state, dev(tau, state), P, packed=True)
    else:
        rbd.mass_matrix_solve_(x, state, dev(tau, state), Mout)
    assert "chol_spec" not in rbd.last_kernel(state) and capfd.readouterr().err == ""


def test_state_kernels_random_trees(rbd, oracle, states_everywhere):
    """Random revolute / prismatic / fixed / sin-cos trees, with and without a 6-dof root, up to the 12 tree levels the kernels keep in registers."""
    from test_chain_plan import random_tree
    rng = np.random.default_rng(19)
    done = 0
    for trial in range(16):
        mech = random_tree(rbd, rng, int(rng.integers(1, 26)), bool(trial % 2), float(rng.uniform(0, 0.6)))
        model = rbd.flatten(mech)
        if states_everywhere == "compiled" and trial % 4:  # a run-time compile is seconds: every fourth tree
            done += 1
            continue
        B, nv = 65, model.nv
        state, q, v, tau, fe = make(rbd, model, B, "f64", "soa", 60 + trial)
        out = torch.zeros_like(state.v)
        rbd.dynamics_bias_(out, state, dev(fe, state))
        ref = oracle.dynamics_bias(model, q, v, fe)
        assert np.abs(host(out, state) - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max()), trial
        result = rbd.DynamicsResult(model, B, layout="soa")
        rbd.mass_matrix_(result, state)
        got = host(result.massmatrix, state).reshape(B, nv, nv).transpose(0, 2, 1)
        Mr = oracle.mass_matrix(model, q)
        il = np.tril_indices(nv)
        assert np.abs(got[:, il[0], il[1]] - Mr[:, il[0], il[1]]).max() <= 1e-9 * max(1.0, np.abs(Mr).max()), trial
        done += 1
    assert done == 16


def test_state_kernels_take_large_batches_by_default(rbd, oracle, models):
    """Without the override: B = 65536 fp32 Atlas (BASELINE configs[2]) goes through the state kernels; whole-batch M and solve checked on a sample."""
    model = models["atlas_floating"]
    B, nv = 65536, model.nv
    state, q, v, tau, _ = make(rbd, model, B, "f32", "aos", 5)
    x = torch.zeros_like(state.v)
    Mout = torch.zeros(B, nv * nv, dtype=torch.float32, device="cuda")
    rbd.mass_matrix_solve_(x, state, dev(tau, state), Mout)
    assert rbd.sync(state) == 0
    assert "chol_spec_f32" in rbd.last_kernel(state) or "crba_state_kernel" in rbd.last_kernel(state)  # (the latter without hiprtc)
    idx = np.arange(0, B, 16)
    Mr = oracle.mass_matrix(model, q[idx], nthreads=8)
    got = Mout[torch.as_tensor(idx, device="cuda")].double().cpu().numpy().reshape(len(idx), nv, nv).transpose(0, 2, 1)
    il = np.tril_indices(nv)
    assert np.abs(got[:, il[0], il[1]] - Mr[:, il[0], il[1]]).max() <= 2e-6 * np.abs(Mr).max()
    Ms, xg = sym(Mr), x[torch.as_tensor(idx, device="cuda")].double().cpu().numpy()
    res = np.einsum("bij,bj->bi", Ms, xg) - tau[idx]
    eta = np.linalg.norm(res, axis=1) / (np.linalg.norm(Ms, axis=(1, 2)) * np.linalg.norm(xg, axis=1) + np.linalg.norm(tau[idx], axis=1))
    assert eta.max() <= 1e-5, eta.max()


def test_compiled_route_is_the_default_and_matches_the_interpreting_one(rbd, oracle, models, monkeypatch):
    """Atlas, fp32, AOS at a batch the state kernels take by default: `mass_matrix!` + Cholesky through the kernels compiled for the mechanism
    (M staged in the factorisation's order, sparse tile Cholesky, M re-emitted as the full square) against the interpreting CRBA + dense
    tile Cholesky on the same inputs, and `mass_matrix!` alone (crba_spec + emit_spec) against the oracle."""
    model = models["atlas_floating"]
    B, nv = 32768 + 7, model.nv  # a ragged last group of 16
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("RBD_JIT", mode)
        state, q, v, tau, _ = make(rbd, model, B, "f32", "aos", 11)
        x = torch.zeros_like(state.v)
        Mout = torch.full((B, nv * nv), float("nan"), dtype=torch.float32, device="cuda")
        rbd.mass_matrix_solve_(x, state, dev(tau, state), Mout)
        assert rbd.sync(state) == 0
        out[mode] = (x.double().cpu().numpy(), Mout.double().cpu().numpy().reshape(B, nv, nv).transpose(0, 2, 1), rbd.last_kernel(state))
        if mode == "1":
            x2 = torch.zeros_like(state.v)
            rbd.mass_matrix_solve_(x2, state, dev(tau, state), None)  # M_out = NULL: no emission
            assert rbd.sync(state) == 0
            assert torch.equal(x2, x)
            M2 = torch.full((B, nv * nv), float("nan"), dtype=torch.float32, device="cuda")
            rbd.mass_matrix_(M2, state)
            assert "emit_spec_f32" in rbd.last_kernel(state)
            assert torch.equal(M2, Mout)  # the same kernels produce it
    if "chol_spec_f32" not in out["1"][2]:
        pytest.skip("hiprtc not available: " + out["1"][2])
    assert "crba_state_kernel" in out["0"][2]
    (xc, Mc, _), (xi, Mi, _) = out["1"], out["0"]
    il = np.tril_indices(nv)
    assert np.isfinite(Mc).all()  # the compiled route writes the whole square
    assert np.abs(Mc - np.transpose(Mc, (0, 2, 1))).max() == 0.0
    scale = np.abs(Mi[:, il[0], il[1]]).max()
    assert np.abs(Mc[:, il[0], il[1]] - Mi[:, il[0], il[1]]).max() <= 2e-6 * scale
    idx = np.arange(0, B, 64)
    Mr = oracle.mass_matrix(model, q[idx], nthreads=8)
    assert np.abs(Mc[idx][:, il[0], il[1]] - Mr[:, il[0], il[1]]).max() <= 2e-6 * np.abs(Mr).max()
    Ms = sym(Mr)
    for xg in (xc, xi):
        res = np.einsum("bij,bj->bi", Ms, xg[idx]) - tau[idx]
        eta = np.linalg.norm(res, axis=1) / (np.linalg.norm(Ms, axis=(1, 2)) * np.linalg.norm(xg[idx], axis=1) + np.linalg.norm(tau[idx], axis=1))
        assert eta.max() <= 1e-5, eta.max()


def test_not_positive_definite_is_reported_by_the_compiled_cholesky(rbd, models, monkeypatch):
    """`rbd_cholesky_solve` keeps the dense kernel; the compiled one sits behind mass_matrix! — a NaN configuration must surface as status 8
    (PosDefException's batched analogue) there too, and must not disturb the other states."""
    tune(monkeypatch, state_min_batch="1")
    model = models["atlas_floating"]
    B = 64
    state, q, v, tau, _ = make(rbd, model, B, "f32", "aos", 12)
    x = torch.zeros_like(state.v)
    rbd.mass_matrix_solve_(x, state, dev(tau, state), None)
    assert rbd.sync(state) == 0
    good = x.clone()
    state.q[5, 9] = float("nan")
    rbd.mass_matrix_solve_(x, state, dev(tau, state), None)
    assert rbd.sync(state) == 8
    keep = [i for i in range(B) if i != 5]
    assert torch.equal(x[keep], good[keep])


# ---- dynamics! compiled for the mechanism (aba_spec, csrc/rbd_spec.hpp) -----------------------------------------------------------------
def backward_error(oracle, model, q, v, tau, fe, vd):
    """‖M v̇ − (τ − c)‖ / (‖M‖‖v̇‖ + ‖τ − c‖) per state: the fp32 measure the other large-batch ABA tests use (the forward error carries cond(M))."""
    Mr = oracle.mass_matrix(model, q, nthreads=8)
    c = oracle.dynamics_bias(model, q, v, fe, nthreads=8)
    Ms = sym(Mr)
    res = np.einsum("bij,bj->bi", Ms, vd) - (tau - c)
    return np.linalg.norm(res, axis=1) / (np.linalg.norm(Ms, axis=(1, 2)) * np.linalg.norm(vd, axis=1) + np.linalg.norm(tau - c, axis=1) + 1e-300)


@pytest.mark.parametrize("layout", ["aos", "soa"])
@pytest.mark.parametrize("name", IN_SCOPE + EVERY_JOINT_TYPE + LIMBS)
def test_compiled_aba_f32(rbd, oracle, models, name, layout):
    """`dynamics!` through the kernel compiled for the mechanism, forced (RBD_ALGO_ABA_COMPILED): a ragged batch (three wavefronts, the last partly
    filled), torques + a wrench on every body + q̇; then no torques and no wrenches; against the oracle (backward error, the q̇ map exactly to fp32)
    and against the lane-per-body kernel on the same inputs."""
    model = models[name]
    B = 150
    state, q, v, tau, fe = make(rbd, model, B, "f32", layout, 71)
    result = rbd.DynamicsResult(model, B, dtype=torch.float32, layout=layout)
    try:
        rbd.dynamics_(result, state, dev(tau, state), dev(fe, state), algorithm="aba_compiled")
    except rbd._capi.RBDError as e:
        if e.status == 3:
            pytest.skip("hiprtc not available")
        raise
    assert rbd.sync(state) == 0 and "aba_spec_f32" in rbd.last_kernel(state)
    vd = host(result.vd, state)
    assert np.isfinite(vd).all()
    assert backward_error(oracle, model, q, v, tau, fe, vd).max() <= 2e-6
    _, qd_ref = oracle.dynamics(model, q, v, tau, fe, want_qdot=True)
    assert np.abs(host(result.qd, state) - qd_ref).max() <= 2e-6 * max(1.0, np.abs(qd_ref).max())
    ref = rbd.DynamicsResult(model, B, dtype=torch.float32, layout=layout)
    rbd.dynamics_(ref, state, dev(tau, state), dev(fe, state), algorithm="aba_lanes")
    assert np.abs(vd - host(ref.vd, state)).max() <= 2e-3 * max(1.0, np.abs(vd).max())  # two fp32 evaluations of an ill-conditioned solve
    result.vd.fill_(float("nan"))
    rbd.dynamics_(result, state, None, None, algorithm="aba_compiled")
    vd = host(result.vd, state)
    assert backward_error(oracle, model, q, v, np.zeros_like(tau), None, vd).max() <= 2e-6


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["aos", "soa"])
@pytest.mark.parametrize("stash", [0, 1])
@pytest.mark.parametrize("name", ["randmech1", "randmech2", "randmech3", "inner_floating", "mixed20"])
def test_compiled_aba_f64_of_mechanisms_no_walk_kernel_takes(rbd, oracle, models, name, layout, stash, monkeypatch):
    """Round 6: `dynamics!` in fp64 for mechanisms with 3-dof joints / 6-dof joints below the world (the reference's own randmech(),
    test/test_mechanism_algorithms.jl:1-11) through the lane-per-state program in doubles (aba_spec_f64): forced (RBD_ALGO_ABA_COMPILED) at a ragged batch and picked
    by the library from its batch threshold on; with a wrench on every body and q̇, without torques and wrenches, and as the M^-1 rhs solve — the reference's 1e-10
    against the oracle.  Atlas-like trees have no such program (the walk kernels are ahead in fp64).  Both programs (csrc/rbd_spec.hpp aba_spec GST): every row in
    LDS (aba_spec_f64), and the spare rows in the workspace's HBM stash (aba_spec_gst_f64: two wavefronts per CU — what the library takes by itself when the
    batch needs fewer of its longer rounds)."""
    model = models[name]
    B = 150
    kernel = "aba_spec_gst_f64" if stash else "aba_spec_f64"
    tune(monkeypatch, spec_f64_stash=stash)
    state, q, v, tau, fe = make(rbd, model, B, "f64", layout, 73)
    result = rbd.DynamicsResult(model, B, layout=layout)
    try:
        rbd.dynamics_(result, state, dev(tau, state), dev(fe, state), algorithm="aba_compiled")
    except rbd._capi.RBDError as e:
        if e.status == 3:
            pytest.skip("hiprtc not available")
        raise
    assert rbd.sync(state) == 0 and kernel in rbd.last_kernel(state)
    ref, qd_ref = oracle.dynamics(model, q, v, tau, fe, want_qdot=True)
    rel = lambda a, b: np.abs(a - b).max() / max(1.0, np.abs(b).max())
    assert rel(host(result.vd, state), ref) <= 1e-10 and rel(host(result.qd, state), qd_ref) <= 1e-13
    result.vd.fill_(float("nan"))
    rbd.dynamics_(result, state, None, None, algorithm="aba_compiled")
    assert rel(host(result.vd, state), oracle.dynamics(model, q, v, None, None)) <= 1e-10
    # the library's own choice from its threshold on (forced down to this batch), and the articulated-body solve M^-1 rhs on the same kernel
    tune(monkeypatch, spec_aba_min_batch=1, spec_f64_stash=stash)
    state2, q2, v2, tau2, _ = make(rbd, model, B, "f64", layout, 74)
    result2 = rbd.DynamicsResult(model, B, layout=layout)
    rbd.dynamics_(result2, state2, dev(tau2, state2))
    assert kernel in rbd.last_kernel(state2), rbd.last_kernel(state2)
    assert rel(host(result2.vd, state2), oracle.dynamics(model, q2, v2, tau2)) <= 1e-10
    x = torch.zeros_like(state2.v)
    rbd.mass_matrix_solve_(x, state2, dev(tau2, state2), algorithm="aba")
    M = oracle.mass_matrix(model, q2)
    xr = np.linalg.solve(sym(M), tau2[..., None])[..., 0]
    assert rel(host(x, state2), xr) <= 1e-9
    assert rbd.jit_source(models["atlas_floating"], torch.float64, "dynamics") is None


@pytest.mark.gpu
def test_compiled_aba_f64_program_by_batch(rbd, oracle, models):
    """Left to itself (no RBD_TUNE) the library takes, of the two fp64 programs, the one whose rounds need less time: one wavefront per CU with every row in LDS,
    two with the stash on a chain 1.7 times as long (1.2 with wrenches) — csrc/rbd_capi.hip run_aba.  Ragged batches of one, two and three chip-fulls of
    wavefronts; parity on the first 1024 states and the ragged tail."""
    model = models["randmech1"]
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    rel = lambda a, b: np.abs(a - b).max() / max(1.0, np.abs(b).max())
    for chips, wrenches, want in ((1, False, "aba_spec_f64"), (2, False, "aba_spec_gst_f64"), (3, False, "aba_spec_f64"), (2, True, "aba_spec_gst_f64"), (3, True, "aba_spec_gst_f64")):
        B = chips * ncu * 64 - 30
        state, q, v, tau, fe = make(rbd, model, B, "f64", "soa", 75 + chips)
        result = rbd.DynamicsResult(model, B, layout="soa")
        try:
            rbd.dynamics_(result, state, dev(tau, state), dev(fe, state) if wrenches else None)
        except rbd._capi.RBDError as e:
            if e.status == 3:
                pytest.skip("hiprtc not available")
            raise
        assert rbd.sync(state) == 0 and want in rbd.last_kernel(state), (chips, wrenches, rbd.last_kernel(state))
        got = host(result.vd, state)
        for sl in (slice(0, 1024), slice(B - 100, B)):
            assert rel(got[sl], oracle.dynamics(model, q[sl], v[sl], tau[sl], fe[sl] if wrenches else None)) <= 1e-10


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,name", [("f32", "double_pendulum"), ("f64", "inner_floating")])
def test_first_use_check_drops_a_wrong_program(rbd, oracle, models, dtype, name, monkeypatch, capfd):
    """The first result a workspace gets from a run-time compiled dynamics! program is compared with the interpreting one-body-per-lane kernel on the call's first
    states (csrc/rbd_capi.hip first_use_check): a program that computes something else — here one that is wrong by construction, RBD_TUNE spec_variant=16: the
    passes on made-up rows instead of q, v, tau — is dropped with a message; asked for by name the call fails, left to the library it is recomputed on the
    interpreting kernels and the caller gets the right v̇.  (Round 6 met a miscompiled variant of the fp64 program in an experiment: profiles/r06_experiments.txt §8.)"""
    model = models[name]
    tune(monkeypatch, spec_variant=16, spec_aba_min_batch=1, state_min_batch=1)
    B = 300
    state, q, v, tau, fe = make(rbd, model, B, dtype, "soa", 77)
    result = rbd.DynamicsResult(model, B, dtype=torch.float32 if dtype == "f32" else torch.float64, layout="soa")
    try:
        rbd.dynamics_(result, state, dev(tau, state), None, algorithm="aba_compiled")
        raise AssertionError("the wrong program was not caught: " + rbd.last_kernel(state))
    except rbd._capi.RBDError as e:
        if e.status == 3 and "differs from the interpreting kernel" not in str(e):
            pytest.skip("hiprtc not available")
        assert e.status == 3 and "differs from the interpreting kernel" in str(e), str(e)
    # the library's own choice: a fresh workspace meets the same wrong program, drops it and recomputes the call
    state2, q2, v2, tau2, _ = make(rbd, model, B, dtype, "soa", 78)
    result2 = rbd.DynamicsResult(model, B, dtype=result.vd.dtype, layout="soa")
    rbd.dynamics_(result2, state2, dev(tau2, state2))
    assert rbd.sync(state2) == 0 and "aba_spec" not in rbd.last_kernel(state2), rbd.last_kernel(state2)
    ref = oracle.dynamics(model, q2, v2, tau2)
    assert np.abs(host(result2.vd, state2) - ref).max() <= (1e-10 if dtype == "f64" else 2e-3) * max(1.0, np.abs(ref).max())
    assert "differs from the interpreting kernel" in capfd.readouterr().err
    # ... and with wrenches, a program of its own (checked on its own first use)
    rbd.dynamics_(result2, state2, dev(tau2, state2), dev(fe, state2))
    assert "aba_spec" not in rbd.last_kernel(state2)
    ref = oracle.dynamics(model, q2, v2, tau2, fe)
    assert np.abs(host(result2.vd, state2) - ref).max() <= (1e-10 if dtype == "f64" else 2e-3) * max(1.0, np.abs(ref).max())


@pytest.mark.gpu
@pytest.mark.parametrize("route,dtype", [("aba_walk", "f64"), ("aba_walk", "f32"), ("aba_banks", "f64"), ("aba_compiled", "f32")])
def test_first_use_check_recomputes_on_the_kernels_built_with_the_library(rbd, oracle, models, route, dtype, monkeypatch, capfd):
    """The same check on the other run-time compiled dynamics! programs — the walk kernel and the two-bodies-per-lane kernel compiled for the mechanism — with
    every check made to find a difference (RBD_TUNE first_use_inject=1): the program is dropped, the call recomputed on the kernel of the same lane mapping that
    was built with the library, the caller sees the right v̇ and a message."""
    model = models["atlas_floating"]
    tune(monkeypatch, first_use_inject=1, walk_min_batch=1, spec_walk_min_batch=1, walk_pair_min_batch=1 << 40, bank_min_batch=1, spec_aba_min_batch=1)
    B = 300
    state, q, v, tau, fe = make(rbd, model, B, dtype, "aos", 79)
    result = rbd.DynamicsResult(model, B, dtype=torch.float32 if dtype == "f32" else torch.float64, layout="aos")
    try:
        rbd.dynamics_(result, state, dev(tau, state), dev(fe, state), algorithm=route)
    except rbd._capi.RBDError as e:
        assert route == "aba_compiled" and e.status == 3 and "first_use_inject" in str(e), str(e)  # (asked for by name: no other kernel may answer)
        return
    assert route != "aba_compiled"
    k = rbd.last_kernel(state)
    assert rbd.sync(state) == 0 and "compiled" not in k and ("walk" in k) == (route == "aba_walk"), k
    assert "first_use_inject" in capfd.readouterr().err
    ref = oracle.dynamics(model, q, v, tau, fe)
    assert np.abs(host(result.vd, state) - ref).max() <= (1e-9 if dtype == "f64" else 5e-3) * max(1.0, np.abs(ref).max())
    # the second call: no check, the kernel built with the library
    rbd.dynamics_(result, state, dev(tau, state), dev(fe, state), algorithm=route)
    assert "compiled" not in rbd.last_kernel(state) and capfd.readouterr().err == ""


@pytest.mark.gpu
@pytest.mark.parametrize("mapping,dtype", [("walk", "f64"), ("banks", "f64"), ("auto", "f32"), ("compiled", "f32")])
def test_first_use_check_of_the_inverse_dynamics_programs(rbd, oracle, models, mapping, dtype, monkeypatch, capfd):
    """... and on the run-time compiled inverse_dynamics! programs (rnea_spec_*, rnea_walk_spec_*, rnea_bank_spec_*) against rnea_kernel: every check made to
    find a difference (RBD_TUNE first_use_inject=1) — dropped, recomputed on the kernel built with the library, right torques, a message."""
    model = models["atlas_floating"]
    tune(monkeypatch, first_use_inject=1, walk_min_batch=1, spec_walk_min_batch=1, rnea_walk_min_batch=1, walk_pair_min_batch=1 << 40, bank_min_batch=1,
         spec_rnea_min_batch=1, state_min_batch=1)
    B = 300
    state, q, v, tau, fe = make(rbd, model, B, dtype, "soa", 80)
    vd = np.random.default_rng(81).standard_normal((B, model.nv))
    out = torch.zeros_like(state.v)
    try:
        rbd.inverse_dynamics_(out, state, dev(vd, state), dev(fe, state), mapping=mapping)
    except rbd._capi.RBDError as e:
        assert mapping == "compiled" and e.status == 3 and "first_use_inject" in str(e), str(e)
        return
    assert mapping != "compiled"
    k = rbd.last_kernel(state)
    assert rbd.sync(state) == 0 and "compiled" not in k, k
    assert "first_use_inject" in capfd.readouterr().err
    ref = oracle.inverse_dynamics(model, q, v, vd, fe)
    assert np.abs(host(out, state) - ref).max() <= (1e-10 if dtype == "f64" else 2e-4) * max(1.0, np.abs(ref).max())
    rbd.inverse_dynamics_(out, state, dev(vd, state), dev(fe, state), mapping=mapping)
    assert "compiled" not in rbd.last_kernel(state) and capfd.readouterr().err == ""


def test_first_call_does_not_wait_for_the_compiler(rbd, models, tmp_path):
    """The library's default (RBD_JIT_ASYNC unset; the test suite otherwise runs with 0): with an EMPTY cache the first `dynamics!` on Atlas at 65 536 fp32
    states returns at once on a kernel that interprets the mechanism while hiprtc compiles `aba_spec_f32` on a background thread (csrc/rbd_jit.hip), a later
    call finds the code object ready and switches to it — same results either way; no call ever waits for the compiler, and a small batch never starts a
    compilation it would not use.  In a process of its own (tests/async_jit_scenario.py), with a hard limit."""
    import json, subprocess, sys
    if rbd.jit_precompile(models["double_pendulum"], torch.float32)[0] is None:
        pytest.skip("hiprtc not available")
    env = dict(os.environ, RBD_JIT_ASYNC="1", RBD_JIT_CACHE=str(tmp_path))
    env.pop("RBD_TUNE", None)
    p = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "async_jit_scenario.py")], env=env, capture_output=True, text=True, timeout=280)
    lines = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
    assert p.returncode == 0 and lines, (p.returncode, p.stdout[-2000:], p.stderr[-2000:])
    r = json.loads(lines[-1][7:])
    assert not r["files_after_small_calls"] and "compiled" not in r["small_kernel"], r  # a 64-state call started no compilation
    assert r["first_call_s"] < 2.0 and "compiled" not in r["first_kernel"] and "aba_spec" not in r["first_kernel"], r
    assert r["slowest_call_s"] < 2.0, r  # no call ever waited for the compiler
    assert "aba_spec_f32" in r["last_kernel"] and any(f.endswith(".hsaco") for f in r["files_at_end"]), r
    assert r["backward_err_first"] <= 2e-6 and r["backward_err_last"] <= 2e-6, r


def test_compiled_aba_random_trees(rbd, oracle):
    """Random revolute / prismatic / fixed / sin-cos trees, with and without a 6-dof root: chains, bushes, branch points below branch points."""
    from test_chain_plan import random_tree
    rng = np.random.default_rng(23)
    done = 0
    for trial in range(6):
        mech = random_tree(rbd, rng, int(rng.integers(2, 22)), bool(trial % 2), float(rng.uniform(0, 0.8)))
        model = rbd.flatten(mech)
        B = 70
        state, q, v, tau, fe = make(rbd, model, B, "f32", "aos", 80 + trial)
        result = rbd.DynamicsResult(model, B, dtype=torch.float32)
        try:
            rbd.dynamics_(result, state, dev(tau, state), dev(fe, state), algorithm="aba_compiled")
        except rbd._capi.RBDError as e:
            if e.status == 3:
                continue  # no hiprtc, or the tree's per-body registers did not fit
            raise
        vd = host(result.vd, state)
        assert backward_error(oracle, model, q, v, tau, fe, vd).max() <= 5e-6, trial
        _, qd_ref = oracle.dynamics(model, q, v, tau, fe, want_qdot=True)
        assert np.abs(host(result.qd, state) - qd_ref).max() <= 2e-6 * max(1.0, np.abs(qd_ref).max()), trial
        done += 1
    assert done >= 4 or done == 0


@pytest.mark.parametrize("layout", ["aos", "soa"])
@pytest.mark.parametrize("name", ["atlas_floating", "double_pendulum", "acrobot_urdf"])
def test_compiled_walk_f64(rbd, oracle, models, name, layout, monkeypatch):
    """The walk kernel compiled for the mechanism (aba_walk_spec, fp64; default from 8192 states up) forced at a small ragged batch: torques + a wrench on
    every body + q̇ at the reference's 1e-10; no torques / no wrenches; a gravity other than the mechanism's."""
    tune(monkeypatch, spec_walk_min_batch="1")
    model = models[name]
    if not rbd.jit_precompile(model, torch.float64)[0]:
        pytest.skip("hiprtc not available")
    B = 150
    state, q, v, tau, fe = make(rbd, model, B, "f64", layout, 44)
    result = rbd.DynamicsResult(model, B, layout=layout)
    rbd.dynamics_(result, state, dev(tau, state), dev(fe, state), algorithm="aba_walk")
    assert rbd.sync(state) == 0
    if os.environ.get("RBD_JIT") == "0":
        assert "aba_walk_kernel" in rbd.last_kernel(state)
    else:
        assert "aba_walk_spec" in rbd.last_kernel(state)
    ref, qd_ref = oracle.dynamics(model, q, v, tau, fe, want_qdot=True)
    assert np.abs(host(result.vd, state) - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max())
    assert np.abs(host(result.qd, state) - qd_ref).max() <= 1e-12 * max(1.0, np.abs(qd_ref).max())
    result.vd.fill_(float("nan"))
    rbd.dynamics_(result, state, None, None, algorithm="aba_walk")
    ref = oracle.dynamics(model, q, v, np.zeros_like(tau), None)
    assert np.abs(host(result.vd, state) - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("layout", ["aos", "soa"])
@pytest.mark.parametrize("name", ["atlas_floating", "double_pendulum", "acrobot_urdf"])
def test_compiled_walk_inverse_dynamics_f64(rbd, oracle, models, name, layout, monkeypatch):
    """rnea_walk_spec (the inverse-dynamics walk kernel compiled for the mechanism, fp64) forced at a small ragged batch: τ with v̇ and wrenches, the per-body
    accelerations and joint wrenches against the oracle's, dynamics_bias!."""
    tune(monkeypatch, spec_walk_min_batch="1")
    model = models[name]
    if not rbd.jit_precompile(model, torch.float64)[0]:
        pytest.skip("hiprtc not available")
    B = 150
    state, q, v, tau, fe = make(rbd, model, B, "f64", layout, 45)
    vd = np.random.default_rng(8).standard_normal((B, model.nv))
    out = torch.zeros_like(state.v)
    shape = (B, 6 * model.n_bodies) if layout == "aos" else (6 * model.n_bodies, B)
    jw = torch.full(shape, float("nan"), dtype=torch.float64, device="cuda"); acc = torch.full_like(jw, float("nan"))
    rbd.inverse_dynamics_(out, state, dev(vd, state), dev(fe, state), mapping="walk", jointwrenchesout=jw, accelerations=acc)
    assert rbd.sync(state) == 0
    assert ("rnea_walk_spec" if os.environ.get("RBD_JIT") != "0" else "rnea_walk_kernel") in rbd.last_kernel(state)
    ref, jw_ref, acc_ref = oracle.inverse_dynamics_bodies(model, q, v, vd, fe)
    assert np.abs(host(out, state) - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max())
    assert np.abs(host(acc, state) - acc_ref.reshape(B, -1)).max() <= 1e-10 * max(1.0, np.abs(acc_ref).max())
    assert np.abs(host(jw, state) - jw_ref.reshape(B, -1)).max() <= 1e-10 * max(1.0, np.abs(jw_ref).max())
    rbd.dynamics_bias_(out, state, mapping="walk")
    ref = oracle.dynamics_bias(model, q, v, None)
    assert np.abs(host(out, state) - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("pair", [False, True])
@pytest.mark.parametrize("name", ["atlas_floating", "double_pendulum", "acrobot_urdf"])
def test_compiled_walk_f32(rbd, oracle, models, name, pair, monkeypatch):
    """The fp32 forms of the compiled walk kernels — one state per lane, and two (packed arithmetic; default from 16 385 states) — forced at a small ragged batch:
    dynamics! by its backward error, q̇, inverse_dynamics! with the per-body outputs at fp32 accuracy."""
    tune(monkeypatch, spec_walk_min_batch="1")
    tune(monkeypatch, walk_pair_min_batch="1" if pair else "1000000000")
    model = models[name]
    B = 333
    state, q, v, tau, fe = make(rbd, model, B, "f32", "aos", 46)
    result = rbd.DynamicsResult(model, B, dtype=torch.float32)
    rbd.dynamics_(result, state, dev(tau, state), dev(fe, state), algorithm="aba_walk")
    assert rbd.sync(state) == 0
    kern = rbd.last_kernel(state)
    if os.environ.get("RBD_JIT") != "0" and "aba_walk_spec" not in kern:
        pytest.skip("hiprtc not available: " + kern)
    assert ("two fp32 states per lane" in kern) == pair
    vd = host(result.vd, state)
    assert np.isfinite(vd).all() and backward_error(oracle, model, q, v, tau, fe, vd).max() <= 2e-6
    _, qd_ref = oracle.dynamics(model, q, v, tau, fe, want_qdot=True)
    assert np.abs(host(result.qd, state) - qd_ref).max() <= 2e-6 * max(1.0, np.abs(qd_ref).max())
    vd_in = np.random.default_rng(9).standard_normal((B, model.nv)).astype(np.float32).astype(np.float64)
    out = torch.zeros_like(state.v)
    jw = torch.full((B, 6 * model.n_bodies), float("nan"), dtype=torch.float32, device="cuda"); acc = torch.full_like(jw, float("nan"))
    rbd.inverse_dynamics_(out, state, dev(vd_in, state), dev(fe, state), mapping="walk", jointwrenchesout=jw, accelerations=acc)
    assert "rnea_walk" in rbd.last_kernel(state) and ("two fp32 states per lane" in rbd.last_kernel(state)) == pair
    ref, jw_ref, acc_ref = oracle.inverse_dynamics_bodies(model, q, v, vd_in, fe)
    tol = lambda r: 2e-4 * max(1.0, np.abs(r).max())
    assert np.abs(host(out, state) - ref).max() <= tol(ref)
    assert np.abs(host(jw, state) - jw_ref.reshape(B, -1)).max() <= tol(jw_ref)
    assert np.abs(host(acc, state) - acc_ref.reshape(B, -1)).max() <= tol(acc_ref)


def test_compiled_walk_random_trees(rbd, oracle, monkeypatch):
    """Random revolute / prismatic / fixed / sin-cos trees with and without a 6-dof root (tests/test_jit_cpu.py compiles the same ones on the CPU)."""
    from test_jit_cpu import walk_trees
    tune(monkeypatch, spec_walk_min_batch="1")
    for trial, model in enumerate(walk_trees(rbd)):
        if not rbd.jit_precompile(model, torch.float64)[0]:
            pytest.skip("hiprtc not available")
        B = 70
        state, q, v, tau, fe = make(rbd, model, B, "f64", "aos", 90 + trial)
        result = rbd.DynamicsResult(model, B)
        rbd.dynamics_(result, state, dev(tau, state), dev(fe, state), algorithm="aba_walk")
        assert "aba_walk_spec" in rbd.last_kernel(state) or os.environ.get("RBD_JIT") == "0"
        ref, qd_ref = oracle.dynamics(model, q, v, tau, fe, want_qdot=True)
        assert np.abs(host(result.vd, state) - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max()), trial
        assert np.abs(host(result.qd, state) - qd_ref).max() <= 1e-12 * max(1.0, np.abs(qd_ref).max()), trial


def test_compiled_walk_is_the_default_for_large_fp64_batches(rbd, oracle, models):
    """16 384 fp64 Atlas states through RBD_ALGO_ABA: aba_walk_spec by default; the whole batch at 1e-10; and a simulate step through it."""
    model = models["atlas_floating"]
    B = 16384
    state, q, v, tau, fe = make(rbd, model, B, "f64", "aos", 47)
    result = rbd.DynamicsResult(model, B)
    rbd.dynamics_(result, state, dev(tau, state), dev(fe, state))
    assert rbd.sync(state) == 0
    if "aba_walk_spec" not in rbd.last_kernel(state):
        pytest.skip("hiprtc not available, or RBD_JIT=0: " + rbd.last_kernel(state))
    ref = oracle.dynamics(model, q, v, tau, fe, nthreads=NT)
    assert np.abs(host(result.vd, state) - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max())


def test_compiled_aba_is_the_default_for_large_fp32_batches(rbd, oracle, models):
    """BASELINE configs[3]'s shard: 65 536 fp32 states go through aba_spec by default; the whole batch's backward error; a NaN state stays alone."""
    model = models["atlas_floating"]
    B = 65536
    state, q, v, tau, _ = make(rbd, model, B, "f32", "aos", 31)
    result = rbd.DynamicsResult(model, B, dtype=torch.float32)
    rbd.dynamics_(result, state, dev(tau, state))
    assert rbd.sync(state) == 0
    if "aba_spec_f32" not in rbd.last_kernel(state):
        pytest.skip("hiprtc not available: " + rbd.last_kernel(state))
    vd = host(result.vd, state)
    assert backward_error(oracle, model, q, v, tau, None, vd).max() <= 2e-6
    good = result.vd.clone()
    state.q[77, 9] = float("nan")
    rbd.dynamics_(result, state, dev(tau, state))
    torch.cuda.synchronize()
    assert not torch.isfinite(result.vd[77]).all()
    keep = torch.ones(B, dtype=torch.bool, device="cuda")
    keep[77] = False
    assert torch.equal(result.vd[keep], good[keep])


# ---- inverse_dynamics! / dynamics_bias! compiled for the mechanism (rnea_spec) ------------------------------------------------------------
@pytest.mark.parametrize("dtype", ["f32", "f64"])
@pytest.mark.parametrize("layout", ["aos", "soa"])
@pytest.mark.parametrize("name", IN_SCOPE + EVERY_JOINT_TYPE + LIMBS)
def test_compiled_rnea(rbd, oracle, models, name, layout, dtype):
    """`inverse_dynamics!` (v̇ and a wrench on every body) and `dynamics_bias!` through rnea_spec, forced, on a ragged batch, against the fp64 oracle
    (fp32: q, v, v̇ staged through LDS; fp64: q alone, v and v̇ read by the lane one body ahead); and against the lane-per-body kernel."""
    model = models[name]
    B = 150
    tol = 2e-5 if dtype == "f32" else 1e-10
    state, q, v, tau, fe = make(rbd, model, B, dtype, layout, 73)
    vd = np.random.default_rng(9).standard_normal((B, model.nv))
    if dtype == "f32":
        vd = vd.astype(np.float32).astype(np.float64)
    out = torch.full_like(state.v, float("nan"))
    try:
        rbd.inverse_dynamics_(out, state, dev(vd, state), dev(fe, state), mapping="compiled")
    except rbd._capi.RBDError as e:
        if e.status == 3:
            pytest.skip("hiprtc not available")
        raise
    assert rbd.sync(state) == 0 and "rnea_spec_" + dtype in rbd.last_kernel(state)
    ref = oracle.inverse_dynamics(model, q, v, vd, fe)
    assert np.abs(host(out, state) - ref).max() <= tol * max(1.0, np.abs(ref).max())
    lanes = torch.zeros_like(state.v)
    rbd.inverse_dynamics_(lanes, state, dev(vd, state), dev(fe, state), mapping="lanes")
    assert np.abs(host(out, state) - host(lanes, state)).max() <= tol * max(1.0, np.abs(ref).max())
    rbd.dynamics_bias_(out, state, mapping="compiled")
    ref = oracle.dynamics_bias(model, q, v, None)
    assert np.abs(host(out, state) - ref).max() <= tol * max(1.0, np.abs(ref).max())


def test_compiled_rnea_is_the_default_for_large_fp32_batches(rbd, oracle, models):
    model = models["atlas_floating"]
    B = 65536
    state, q, v, tau, fe = make(rbd, model, B, "f32", "aos", 32)
    vd = np.random.default_rng(10).standard_normal((B, model.nv)).astype(np.float32).astype(np.float64)
    out = torch.zeros_like(state.v)
    rbd.inverse_dynamics_(out, state, dev(vd, state), dev(fe, state))
    assert rbd.sync(state) == 0
    if "rnea_spec_f32" not in rbd.last_kernel(state):
        pytest.skip("hiprtc not available: " + rbd.last_kernel(state))
    ref = oracle.inverse_dynamics(model, q, v, vd, fe, nthreads=8)
    assert np.abs(host(out, state) - ref).max() <= 2e-4 * max(1.0, np.abs(ref).max())


def test_compiled_aba_as_the_mass_matrix_solve(rbd, oracle, models):
    """M⁻¹ rhs by the articulated-body pass (v = 0, g = 0, τ = rhs) at a batch the compiled kernel takes: gravity is a kernel argument there, not the
    plan's constant.  Backward error of M x = rhs against the oracle's M."""
    model = models["atlas_floating"]
    B = 40000
    state, q, v, tau, _ = make(rbd, model, B, "f32", "aos", 33)
    x = torch.zeros_like(state.v)
    rbd.mass_matrix_solve_(x, state, dev(tau, state), algorithm="aba")
    assert rbd.sync(state) == 0
    if "aba_spec_f32" not in rbd.last_kernel(state):
        pytest.skip("hiprtc not available: " + rbd.last_kernel(state))
    idx = np.arange(0, B, 40)
    Ms = sym(oracle.mass_matrix(model, q[idx], nthreads=8))
    xg = x[torch.as_tensor(idx, device="cuda")].double().cpu().numpy()
    res = np.einsum("bij,bj->bi", Ms, xg) - tau[idx]
    eta = np.linalg.norm(res, axis=1) / (np.linalg.norm(Ms, axis=(1, 2)) * np.linalg.norm(xg, axis=1) + np.linalg.norm(tau[idx], axis=1))
    assert eta.max() <= 2e-6, eta.max()


@pytest.mark.parametrize("dtype", ["f32", "f64"])
@pytest.mark.parametrize("name", EVERY_JOINT_TYPE + LIMBS)
def test_compiled_mass_matrix_every_joint_type(rbd, oracle, models, name, dtype, monkeypatch):
    """`mass_matrix!` of mechanisms with 3-dof joints and 6-dof joints below the world through crba_spec (batch-innermost M: the kernel's own stores), a ragged
    batch, structural zeros included; and the same M through the Cholesky solve."""
    tune(monkeypatch, state_min_batch="1")
    model = models[name]
    B, nv = 70, model.nv
    state, q, v, tau, _ = make(rbd, model, B, dtype, "soa", 43)
    result = rbd.DynamicsResult(model, B, dtype=TD[dtype], layout="soa")
    result.massmatrix.fill_(float("nan"))
    rbd.mass_matrix_(result, state)
    if "crba_spec" not in rbd.last_kernel(state):
        pytest.skip("hiprtc not available: " + rbd.last_kernel(state))
    got = host(result.massmatrix, state).reshape(B, nv, nv).transpose(0, 2, 1)
    Mr = oracle.mass_matrix(model, q)
    il = np.tril_indices(nv)
    assert np.isfinite(got[:, il[0], il[1]]).all()
    assert np.abs(got[:, il[0], il[1]] - Mr[:, il[0], il[1]]).max() <= (1e-10 if dtype == "f64" else 2e-6) * max(1.0, np.abs(Mr).max())
    x = torch.zeros_like(state.v)
    rbd.mass_matrix_solve_(x, state, dev(tau, state), result.massmatrix)
    assert rbd.sync(state) == 0
    Ms = sym(Mr)
    xg = host(x, state)
    res = np.einsum("bij,bj->bi", Ms, xg) - tau
    eta = np.linalg.norm(res, axis=1) / (np.linalg.norm(Ms, axis=(1, 2)) * np.linalg.norm(xg, axis=1) + np.linalg.norm(tau, axis=1))
    assert eta.max() <= (1e-12 if dtype == "f64" else 1e-5), eta.max()


@pytest.mark.parametrize("layout", ["aos", "soa"])
@pytest.mark.parametrize("dtype,name", [("f32", "atlas_floating"), ("f32", "limbs_humanoid"), ("f32", "mixed20"), ("f32", "double_pendulum"), ("f64", "atlas_floating"),
                                        ("f64", "randmech1")])
def test_mass_matrix_solve_packed(rbd, oracle, models, name, dtype, layout, states_everywhere):
    """`rbd_mass_matrix_solve_packed`: M as LAPACK's packed lower triangle (what the reference's Symmetric(:L) defines) — from the tile Cholesky's own tiles
    (fp32 with compiled kernels, state-major) or packed from the square (every other route); a ragged batch; x as in the square form."""
    model = models[name]
    B, nv = 150, model.nv
    state, q, v, tau, _ = make(rbd, model, B, dtype, layout, 91)
    np_ = nv * (nv + 1) // 2
    x = torch.zeros_like(state.v)
    # (a guard behind the buffer: the kernel writes a state's run in 16-byte pieces that straddle states — none may land past the last state)
    whole = torch.full((B * np_ + 64,), float("nan"), dtype=TD[dtype], device="cuda")
    P = whole[:B * np_].view((B, np_) if layout == "aos" else (np_, B))
    rbd.mass_matrix_solve_(x, state, dev(tau, state), P, packed=True)
    assert rbd.sync(state) == 0
    assert bool(torch.isnan(whole[B * np_:]).all())
    if dtype == "f32" and layout == "aos" and states_everywhere == "compiled" and nv % 4 == 0 and nv <= 40:
        assert "chol_spec_packed_f32" in rbd.last_kernel(state), rbd.last_kernel(state)
    got = rbd.unpack_lower(host(P, state), nv)
    Mr = oracle.mass_matrix(model, q)
    il = np.tril_indices(nv)
    assert np.isfinite(got[:, il[0], il[1]]).all()
    assert np.abs(got[:, il[0], il[1]] - Mr[:, il[0], il[1]]).max() <= (1e-10 if dtype == "f64" else 2e-6) * max(1.0, np.abs(Mr).max())
    Ms, xg = sym(Mr), host(x, state)
    res = np.einsum("bij,bj->bi", Ms, xg) - tau
    eta = np.linalg.norm(res, axis=1) / (np.linalg.norm(Ms, axis=(1, 2)) * np.linalg.norm(xg, axis=1) + np.linalg.norm(tau, axis=1))
    assert eta.max() <= (1e-12 if dtype == "f64" else 1e-5), eta.max()
