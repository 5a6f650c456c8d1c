"""GPU parity: the HIP path (through the C ABI of librbd_hip.so) against the CPU oracle on identical seeded inputs.

Tolerances (stated per BASELINE.md §4):
  fp64: max |x_gpu - x_oracle| <= 1e-10 * max(1, max|x_oracle|)   (the reference's own atol is 1e-10,
        test/test_mechanism_algorithms.jl:739; GPU ABA vs the oracle's CRBA+Cholesky route differ by ~1e-13 rel.)
  fp32: v̇ is cond(M)-limited (cond ≈ 5e5 on Atlas) so parity is stated as backward error
        ||M v̇ - (τ - c)|| / ||τ - c|| <= 2e-5 with M, c from the fp64 oracle, plus the forward bound 8·cond₂(M_b)·eps32 state by state
        (precedent atol 1e-3, test/test_mechanism_modification.jl:339);
        τ (RNEA) and M (CRBA) have no solve: relative 2e-5.
"""
import os

import numpy as np
import pytest
import torch

from conftest import rand_inputs, tune

pytestmark = pytest.mark.gpu

MODELS = ["atlas_floating", "atlas_fixed", "valkyrie_floating", "double_pendulum", "acrobot_urdf", "randmech1", "randmech2", "randmech3", "inner_floating"]
TD = {"f64": torch.float64, "f32": torch.float32}
NT = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 1  # host threads for full-batch oracle runs


def assert_fp32_forward(oracle, model, q, got, ref, C=8.0, what=""):
    """Forward error of an fp32 solve of M v̇ = τ − c, state by state: ||got − ref|| / ||ref|| <= C · cond₂(M_b) · eps32 — what a backward-stable
    fp32 algorithm on fp32-rounded inputs can deliver for THAT state's mass matrix (round-1 review: not a blanket constant)."""
    M = oracle.mass_matrix(model, q, nthreads=NT)
    Ms = np.tril(M) + np.transpose(np.tril(M, -1), (0, 2, 1))
    kappa = np.linalg.cond(Ms)
    err = np.linalg.norm(got - ref, axis=1) / np.maximum(np.linalg.norm(ref, axis=1), 1e-30)
    bound = C * kappa * np.finfo(np.float32).eps
    worst = int(np.argmax(err / bound))
    assert (err <= bound).all(), (what, worst, float(err[worst]), float(bound[worst]), float(kappa[worst]))

ND = {"f64": np.float64, "f32": np.float32}


def dev(x, state):
    t = torch.as_tensor(np.ascontiguousarray(x), dtype=state.dtype)
    if state.layout == "soa":
        t = t.t().contiguous()
    return t.cuda()


def host(t, state):
    t = t.detach().cpu()
    if state.layout == "soa":
        t = t.t()
    return t.double().numpy()


def make(rbd, model, B, dtype, layout, seed, fext=True):
    q, v, tau, fe = rand_inputs(rbd, model, B, seed, fext=True)
    state = rbd.MechanismState(model, B, dtype=TD[dtype], layout=layout)
    rbd.set_configuration_(state, q)
    rbd.set_velocity_(state, v)
    # inputs as seen by the device (fp32 rounding of inputs is not the kernel's error)
    q, v, tau, fe = [a.astype(ND[dtype]).astype(np.float64) for a in (q, v, tau, fe)]
    return state, q, v, tau, (fe if fext else None)


@pytest.mark.parametrize("layout", ["aos", "soa"])
@pytest.mark.parametrize("name", MODELS)
def test_dynamics_f64(rbd, oracle, models, name, layout):
    model = models[name]
    B = 67  # ragged: not a multiple of the states-per-wave of any model
    state, q, v, tau, fe = make(rbd, model, B, "f64", layout, 11)
    result = rbd.DynamicsResult(model, B, layout=layout)
    rbd.dynamics_(result, state, dev(tau, state), dev(fe, state))
    torch.cuda.synchronize()
    ref, qd_ref = oracle.dynamics(model, q, v, tau, fe, want_qdot=True)
    got = host(result.vd, state)
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max())
    assert np.abs(host(result.qd, state) - qd_ref).max() <= 1e-13 * max(1.0, np.abs(qd_ref).max())


@pytest.mark.parametrize("name", MODELS)
def test_dynamics_defaults_no_tau_no_wrenches(rbd, oracle, models, name):
    model = models[name]
    B = 5
    state, q, v, tau, fe = make(rbd, model, B, "f64", "aos", 12)
    result = rbd.DynamicsResult(model, B)
    rbd.dynamics_(result, state)  # torques = ConstVector(0), externalwrenches = NullDict
    ref = oracle.dynamics(model, q, v)
    assert np.abs(host(result.vd, state) - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("layout", ["aos", "soa"])
@pytest.mark.parametrize("name", MODELS)
def test_inverse_dynamics_and_bias_f64(rbd, oracle, models, name, layout):
    model = models[name]
    B = 33
    state, q, v, vd, fe = make(rbd, model, B, "f64", layout, 13)
    out = torch.zeros_like(state.v)
    rbd.inverse_dynamics_(out, state, dev(vd, state), dev(fe, state))
    ref = oracle.inverse_dynamics(model, q, v, vd, fe)
    assert np.abs(host(out, state) - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max())
    result = rbd.DynamicsResult(model, B, layout=layout)
    rbd.dynamics_bias_(result, state, dev(fe, state))
    ref = oracle.dynamics_bias(model, q, v, fe)
    assert np.abs(host(result.dynamicsbias, state) - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max())
    rbd.dynamics_bias_(result, state)
    ref = oracle.dynamics_bias(model, q, v)
    assert np.abs(host(result.dynamicsbias, state) - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("layout", ["aos", "soa"])
@pytest.mark.parametrize("name", MODELS)
def test_mass_matrix_f64(rbd, oracle, models, name, layout):
    model = models[name]
    B = 19
    state, q, v, _, _ = make(rbd, model, B, "f64", layout, 14)
    result = rbd.DynamicsResult(model, B, layout=layout)
    result.massmatrix.fill_(float("nan"))  # the lower triangle must be fully written, zeros included
    rbd.mass_matrix_(result, state)
    nv = model.nv
    got = host(result.massmatrix, state).reshape(B, nv, nv).transpose(0, 2, 1)  # [b, i, j]
    ref = oracle.mass_matrix(model, q)
    il = np.tril_indices(nv)
    g, r = got[:, il[0], il[1]], ref[:, il[0], il[1]]
    assert np.isfinite(g).all()
    assert np.abs(g - r).max() <= 1e-10 * max(1.0, np.abs(r).max())
    # structural zeros (dof j does not support body(i): support_set_masks) are exact zeros, like the reference
    # (mechanism_algorithms.jl:263-267)
    body = np.zeros(nv, dtype=int)
    from rigidbodydynamics_jl_amd.mechanism import _NV
    for i in range(model.n_bodies):
        body[model.v_offset[i]:model.v_offset[i] + _NV[int(model.joint_type[i])]] = i
    def supports(j, i):
        b = body[i]
        while b >= 0:
            if b == body[j]:
                return True
            b = model.parent[b]
        return False
    unsupported = np.array([not supports(j, i) for i, j in zip(*il)])
    assert (g[:, unsupported] == 0).all() and (r[:, unsupported] == 0).all()


@pytest.mark.parametrize("name", ["atlas_floating", "atlas_fixed", "double_pendulum"])
def test_dynamics_f32(rbd, oracle, models, name):
    model = models[name]
    B = 257
    state, q, v, tau, fe = make(rbd, model, B, "f32", "aos", 15)
    result = rbd.DynamicsResult(model, B, dtype=torch.float32)
    rbd.dynamics_(result, state, dev(tau, state), dev(fe, state))
    got = host(result.vd, state)
    ref = oracle.dynamics(model, q, v, tau, fe)
    back = oracle.inverse_dynamics(model, q, v, got, fe)
    c = oracle.dynamics_bias(model, q, v, fe)
    rel = np.linalg.norm(back - tau, axis=1) / np.linalg.norm(tau - c, axis=1)
    assert rel.max() <= 2e-5, rel.max()
    assert_fp32_forward(oracle, model, q, got, ref, what=name)


@pytest.mark.parametrize("name", ["atlas_floating", "double_pendulum"])
def test_rnea_crba_f32(rbd, oracle, models, name):
    model = models[name]
    B = 64
    state, q, v, vd, fe = make(rbd, model, B, "f32", "aos", 16)
    out = torch.zeros_like(state.v)
    rbd.inverse_dynamics_(out, state, dev(vd, state), dev(fe, state))
    ref = oracle.inverse_dynamics(model, q, v, vd, fe)
    assert np.abs(host(out, state) - ref).max() <= 2e-5 * np.abs(ref).max()
    result = rbd.DynamicsResult(model, B, dtype=torch.float32)
    rbd.mass_matrix_(result, state)
    nv = model.nv
    got = host(result.massmatrix, state).reshape(B, nv, nv).transpose(0, 2, 1)
    ref = oracle.mass_matrix(model, q)
    il = np.tril_indices(nv)
    assert np.abs(got[:, il[0], il[1]] - ref[:, il[0], il[1]]).max() <= 2e-5 * np.abs(ref).max()


@pytest.mark.parametrize("B", [1, 2, 3, 4096])
def test_batch_sizes_and_round_trip(rbd, oracle, models, B):
    """Edge batch sizes; at the BASELINE size (4096) also the size-independent property of
    test/test_mechanism_algorithms.jl:729-740: dynamics! then inverse_dynamics! returns τ (GPU-only round trip)."""
    model = models["atlas_floating"]
    state, q, v, tau, fe = make(rbd, model, B, "f64", "aos", 17)
    result = rbd.DynamicsResult(model, B)
    t, f = dev(tau, state), dev(fe, state)
    rbd.dynamics_(result, state, t, f)
    back = torch.zeros_like(state.v)
    rbd.inverse_dynamics_(back, state, result.vd, f)
    c = torch.zeros_like(state.v)
    rbd.dynamics_bias_(c, state, f)
    scale = float((t - c).abs().max())
    assert float((back - t).abs().max()) <= 1e-10 * max(1.0, scale)
    ref = oracle.dynamics(model, q, v, tau, fe, nthreads=NT)  # every state of the batch, also at the BASELINE size
    assert np.abs(host(result.vd, state) - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max())


def test_kinetic_energy_identity_full_size(rbd, models):
    """½ v'Mv computed from the GPU mass matrix equals v·(ID(v̇=v) - ID(0))/2 … i.e. M v = ID(q, 0-velocity, v̇=v) - ID(q,0,0):
    a CPU-free consistency check of CRBA against RNEA at B = 4096."""
    model = models["atlas_floating"]
    B = 4096
    state, q, v, _, _ = make(rbd, model, B, "f64", "aos", 18)
    result = rbd.DynamicsResult(model, B)
    rbd.mass_matrix_(result, state)
    M = result.massmatrix_dense()
    vv = state.v.clone()
    rbd.set_velocity_(state, torch.zeros_like(vv))
    a, g = torch.zeros_like(vv), torch.zeros_like(vv)
    rbd.inverse_dynamics_(a, state, vv)
    rbd.dynamics_bias_(g, state)
    Mv = torch.einsum("bij,bj->bi", M, vv)
    assert float((Mv - (a - g)).abs().max()) <= 1e-9 * float(Mv.abs().max())


def test_errors(rbd, models):
    model = models["atlas_floating"]
    state = rbd.MechanismState(model, 4)
    result = rbd.DynamicsResult(model, 4)
    with pytest.raises(rbd.DimensionMismatch):
        rbd.dynamics_(result, state, torch.zeros(4, model.nv + 1, dtype=torch.float64, device="cuda"))
    with pytest.raises(rbd.DimensionMismatch):
        rbd.mass_matrix_(torch.zeros(4, 5, dtype=torch.float64, device="cuda"), state)
    with pytest.raises(ValueError):
        rbd.dynamics_(result, state, torch.zeros(4, model.nv, dtype=torch.float32, device="cuda"))


# ---- the reference's own route: dynamics_bias! + mass_matrix! + potrf!/potrs! (RBD_ALGO_CRBA_CHOLESKY) ---------------
@pytest.mark.parametrize("layout", ["aos", "soa"])
@pytest.mark.parametrize("name", ["atlas_floating", "atlas_fixed", "double_pendulum", "acrobot_urdf"])
def test_dynamics_crba_cholesky_route_f64(rbd, oracle, models, name, layout):
    model = models[name]
    B = 37
    state, q, v, tau, fe = make(rbd, model, B, "f64", layout, 21)
    result = rbd.DynamicsResult(model, B, layout=layout)
    rbd.dynamics_(result, state, dev(tau, state), dev(fe, state), algorithm="crba")
    ref, qd_ref = oracle.dynamics(model, q, v, tau, fe, want_qdot=True)
    got = host(result.vd, state)
    assert np.abs(got - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max())
    assert np.abs(host(result.qd, state) - qd_ref).max() <= 1e-13 * max(1.0, np.abs(qd_ref).max())
    # the DynamicsResult side products of the reference: massmatrix (lower) and dynamicsbias
    nv = model.nv
    Mg = host(result.massmatrix, state).reshape(B, nv, nv).transpose(0, 2, 1)
    Mr = oracle.mass_matrix(model, q)
    il = np.tril_indices(nv)
    assert np.abs(Mg[:, il[0], il[1]] - Mr[:, il[0], il[1]]).max() <= 1e-10 * max(1.0, np.abs(Mr).max())
    cr = oracle.dynamics_bias(model, q, v, fe)
    assert np.abs(host(result.dynamicsbias, state) - cr).max() <= 1e-10 * max(1.0, np.abs(cr).max())
    # and it agrees with the fused ABA kernel
    r2 = rbd.DynamicsResult(model, B, layout=layout)
    rbd.dynamics_(r2, state, dev(tau, state), dev(fe, state), algorithm="aba")
    assert float((r2.vd - result.vd).abs().max()) <= 1e-10 * max(1.0, np.abs(ref).max())


def test_mass_matrix_solve_f32_config3(rbd, oracle, models):
    """BASELINE configs[2] shape at a size the oracle finishes quickly: Atlas floating fp32 mass_matrix! + Cholesky solve.
    Backward error ||M x - rhs|| / ||rhs|| with the fp64 oracle's M (cond(M) ~ 5e5 limits the forward error)."""
    model = models["atlas_floating"]
    B = 1024
    state, q, v, tau, _ = make(rbd, model, B, "f32", "aos", 22)
    x = torch.zeros_like(state.v)
    Mout = torch.zeros(B, model.nv * model.nv, dtype=torch.float32, device="cuda")
    rbd.mass_matrix_solve_(x, state, dev(tau, state), Mout)
    assert rbd.sync(state) == 0
    M = oracle.mass_matrix(model, q)
    Ms = np.tril(M) + np.transpose(np.tril(M, -1), (0, 2, 1))
    xg = host(x, state)
    # normwise backward error (Rigal–Gaches): ||M x - r|| / (||M|| ||x|| + ||r||)  — fp32 Cholesky: O(n eps) ~ 2e-6
    res = np.einsum("bij,bj->bi", Ms, xg) - tau
    eta = np.linalg.norm(res, axis=1) / (np.linalg.norm(Ms, axis=(1, 2)) * np.linalg.norm(xg, axis=1) + np.linalg.norm(tau, axis=1))
    assert eta.max() <= 1e-5, eta.max()
    xr = np.linalg.solve(Ms, tau[..., None])[..., 0]
    assert_fp32_forward(oracle, model, q, xg, xr, what="config3")


def test_mass_matrix_solve_full_size_property(rbd, oracle, models):
    """configs[2] at full size (B = 65536 fp32): the GPU's M against the oracle's on 8192 states spread over the batch and x through
    the backward error with the ORACLE's M there (a wrong-but-SPD M cannot pass), plus the size-independent property M (M^-1 r) = r
    over the whole batch (torch fp64 matmul as the checker)."""
    model = models["atlas_floating"]
    B = 65536
    state, q, v, tau, _ = make(rbd, model, B, "f32", "aos", 23)
    rhs = dev(tau, state)
    x = torch.zeros_like(state.v)
    Mout = torch.zeros(B, model.nv * model.nv, dtype=torch.float32, device="cuda")
    rbd.mass_matrix_solve_(x, state, rhs, Mout)
    nv = model.nv
    L = torch.tril(Mout.reshape(B, nv, nv).transpose(1, 2).double())
    Ms = L + torch.tril(L, -1).transpose(1, 2)
    res = torch.einsum("bij,bj->bi", Ms, x.double()) - rhs.double()
    eta = res.norm(dim=1) / (torch.linalg.matrix_norm(Ms) * x.double().norm(dim=1) + rhs.double().norm(dim=1))
    assert float(eta.max()) <= 1e-5
    idx = np.arange(0, B, 8)  # 8192 states, every wavefront of the launch represented
    Mo = oracle.mass_matrix(model, q[idx], nthreads=NT)
    Mo = np.tril(Mo) + np.transpose(np.tril(Mo, -1), (0, 2, 1))
    Mg = Ms[torch.as_tensor(idx, device=Ms.device)].cpu().numpy()
    assert np.abs(Mg - Mo).max() <= 2e-5 * np.abs(Mo).max()  # fp32 CRBA, no solve involved
    xg = x[torch.as_tensor(idx, device=x.device)].double().cpu().numpy()
    r = tau[idx]
    res = np.einsum("bij,bj->bi", Mo, xg) - r
    eta = np.linalg.norm(res, axis=1) / (np.linalg.norm(Mo, axis=(1, 2)) * np.linalg.norm(xg, axis=1) + np.linalg.norm(r, axis=1))
    assert eta.max() <= 1e-5


def test_not_positive_definite_is_reported(rbd, models):
    """potrf! throws PosDefException in the reference (src/mechanism_algorithms.jl:764); here rbd_sync reports status 8."""
    model = rbd.flatten(rbd.double_pendulum())
    model.inertia_moment = -model.inertia_moment  # unphysical: M is not SPD
    model.inertia_mass = -model.inertia_mass
    model.inertia_cross = -model.inertia_cross
    model._c = None
    state = rbd.MechanismState(model, 4)
    result = rbd.DynamicsResult(model, 4)
    rbd.rand_(state, 3)
    rbd.dynamics_(result, state, algorithm="crba")
    assert rbd.sync(state) == 8
    assert rbd.sync(state) == 0  # the flag is cleared once reported


# ---- loop joints: BASELINE configs[4] — four-bar linkage (test/test_simulate.jl:127-190), B = 4096 fp64 -------------------
def four_bar_inputs(rbd, B, seed):
    rng = np.random.default_rng(seed)
    q = np.tile(rbd.FOUR_BAR_INITIAL_Q, (B, 1))
    q[:, 0] += rng.uniform(-0.05, 0.05, B)  # constraint violation exercises the Baumgarte term (SURVEY.md §8 d, config 5)
    v = np.tile(rbd.FOUR_BAR_INITIAL_V, (B, 1))
    tau = rng.random((B, 3))
    return q, v, tau


@pytest.mark.parametrize("kernels", ["compiled", "generic"])
@pytest.mark.parametrize("layout", ["aos", "soa"])
@pytest.mark.parametrize("stabilize", [True, False])
def test_four_bar_dynamics_f64(rbd, oracle, models, stabilize, layout, kernels, monkeypatch):
    """kernels: the loop branch as straight-line code compiled for the mechanism at run time (csrc/rbd_loop_small.hpp against constant tables through
    rbd_jit.hip — the default where hiprtc is available), or the same code reading the tables from device memory (RBD_JIT=0: loop_fused_small_kernel)."""
    monkeypatch.setenv("RBD_JIT", "1" if kernels == "compiled" else "0")
    model = models["four_bar"]
    B = 4096
    q, v, tau = four_bar_inputs(rbd, B, 31)
    state = rbd.MechanismState(model, B, layout=layout)
    result = rbd.DynamicsResult(model, B, layout=layout)
    rbd.set_configuration_(state, q)
    rbd.set_velocity_(state, v)
    rbd.dynamics_(result, state, dev(tau, state), stabilization_gains="default" if stabilize else None)
    assert rbd.sync(state) == 0
    assert ("loop_spec" in rbd.last_kernel(state)) == (kernels == "compiled") or kernels == "compiled"  # (no hiprtc: the generic kernel serves both)
    n = 256
    ref = oracle.dynamics_loops(model, q[:n], v[:n], tau[:n], stabilize=stabilize)
    got = host(result.vd, state)
    assert np.abs(got[:n] - ref["vdot"]).max() <= 1e-10 * max(1.0, np.abs(ref["vdot"]).max())
    nv, nc = model.nv, model.nc
    K = host(result.constraintjacobian, state).reshape(B, nv, nc).transpose(0, 2, 1)  # nc×nv column-major per state -> [b, c, v]
    k = host(result.constraintbias, state)
    lam = host(result.lambda_, state)
    assert np.abs(K[:n] - ref["K"]).max() <= 1e-12
    assert np.abs(k[:n] - ref["k"]).max() <= 1e-10 * max(1.0, np.abs(ref["k"]).max())
    # λ: both sides take the minimum-norm solution (gelsy! semantics); v̇ does not depend on the choice
    assert np.abs(lam[:n] - ref["lam"]).max() <= 1e-8 * max(1.0, np.abs(ref["lam"]).max())
    # size-independent properties at the full batch: the KKT conditions of dynamics! (mechanism_algorithms.jl:828-836)
    Mg = host(result.massmatrix, state).reshape(B, nv, nv).transpose(0, 2, 1)
    Ms = np.tril(Mg) + np.transpose(np.tril(Mg, -1), (0, 2, 1))
    c = host(result.dynamicsbias, state)
    r1 = np.einsum("bij,bj->bi", Ms, got) + c + np.einsum("bcv,bc->bv", K, lam) - tau
    assert np.abs(r1).max() <= 1e-10
    r2 = np.einsum("bcv,bv->bc", K, got) + k
    assert np.abs(r2).max() <= 1e-8


def test_inverse_dynamics_rejects_loops(rbd, models):
    """inverse_dynamics! 'can currently only handle tree Mechanisms' (src/mechanism_algorithms.jl:549)."""
    model = models["four_bar"]
    state = rbd.MechanismState(model, 4)
    with pytest.raises(RuntimeError, match="tree Mechanisms"):
        rbd.inverse_dynamics_(torch.zeros_like(state.v), state, torch.zeros_like(state.v))


# ---- the C ABI driven directly (what the Julia shim does): host buffers (RBD_MEM_HOST), timing hooks, B = 0 ----------------
def test_c_abi_host_memory_mode(rbd, oracle, models):
    import ctypes
    from rigidbodydynamics_jl_amd import _capi
    from rigidbodydynamics_jl_amd.state import _Model
    model = models["atlas_floating"]
    B = 33
    q, v, tau, fe = rand_inputs(rbd, model, B, 41, fext=True)
    L = _capi.lib()
    m = _Model(model)
    ws = ctypes.c_void_p()
    assert L.rbd_workspace_create(m.handle, B, 0, _capi.F64, None, ctypes.byref(ws)) == 0
    opts = _capi.Opts(_capi.LAYOUT_AOS, _capi.MEM_HOST, _capi.ALGO_ABA, 1)
    vd, qd = np.zeros((B, model.nv)), np.zeros((B, model.nq))
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    assert L.rbd_workspace_enable_timing(ws, 1) == 0
    assert L.rbd_dynamics(ws, B, P(q), P(v), P(tau), P(fe), P(vd), P(qd), None, ctypes.byref(opts)) == 0
    assert L.rbd_sync(ws) == 0
    ms = ctypes.c_float()
    assert L.rbd_workspace_last_kernel_ms(ws, ctypes.byref(ms)) == 0 and 0 < ms.value < 50
    ref, qd_ref = oracle.dynamics(model, q, v, tau, fe, want_qdot=True)
    assert np.abs(vd - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max())
    assert np.abs(qd - qd_ref).max() <= 1e-13 * max(1.0, np.abs(qd_ref).max())
    # inverse dynamics, bias and mass matrix through host buffers
    t = np.zeros((B, model.nv))
    assert L.rbd_inverse_dynamics(ws, B, P(q), P(v), P(vd), P(fe), P(t), ctypes.byref(opts)) == 0
    assert L.rbd_sync(ws) == 0
    assert np.abs(t - tau).max() <= 1e-9 * max(1.0, np.abs(tau).max())   # round trip dynamics! -> inverse_dynamics!
    Mh = np.full((B, model.nv * model.nv), np.nan)
    assert L.rbd_mass_matrix(ws, B, P(q), P(Mh), ctypes.byref(opts)) == 0
    assert L.rbd_sync(ws) == 0
    Mr = oracle.mass_matrix(model, q)
    il = np.tril_indices(model.nv)
    got = Mh.reshape(B, model.nv, model.nv).transpose(0, 2, 1)
    assert np.abs(got[:, il[0], il[1]] - Mr[:, il[0], il[1]]).max() <= 1e-10 * np.abs(Mr).max()
    # kinematics by-products through host buffers
    A, com, en, mom, J = (np.zeros((B, 6 * model.nv)), np.zeros((B, 3)), np.zeros((B, 2)), np.zeros((B, 12)), np.zeros((B, 6 * model.nv)))
    assert L.rbd_kinematics(ws, B, P(q), P(v), P(A), P(com), P(en), ctypes.byref(opts)) == 0
    assert L.rbd_momentum(ws, B, P(q), P(v), P(mom), ctypes.byref(opts)) == 0
    assert L.rbd_geometric_jacobian(ws, B, P(q), -1, model.n_bodies - 1, P(J), ctypes.byref(opts)) == 0
    assert L.rbd_sync(ws) == 0
    A_ref, h_ref, com_ref = oracle.momentum_matrix(model, q, v)
    ke_ref, pe_ref = oracle.energy(model, q, v)
    J_ref, _ = oracle.geometric_jacobian(model, q, -1, model.n_bodies - 1)
    assert np.abs(A.reshape(B, model.nv, 6).transpose(0, 2, 1) - A_ref).max() <= 1e-11 * np.abs(A_ref).max()
    assert np.abs(com - com_ref).max() <= 1e-12 and np.abs(en[:, 0] - ke_ref).max() <= 1e-11 * np.abs(ke_ref).max()
    assert np.abs(en[:, 1] - pe_ref).max() <= 1e-11 * np.abs(pe_ref).max()
    assert np.abs(mom[:, :6] - h_ref).max() <= 1e-11 * np.abs(h_ref).max()
    assert np.abs(J.reshape(B, model.nv, 6).transpose(0, 2, 1) - J_ref).max() <= 1e-12
    # B = 0 is a no-op; B > max_batch is a DimensionMismatch; null q is an ArgumentError
    assert L.rbd_dynamics(ws, 0, P(q), P(v), None, None, P(vd), None, None, ctypes.byref(opts)) == 0
    assert L.rbd_dynamics(ws, B + 1, P(q), P(v), None, None, P(vd), None, None, ctypes.byref(opts)) == 2
    assert L.rbd_dynamics(ws, B, None, P(v), None, None, P(vd), None, None, ctypes.byref(opts)) == 1
    assert L.rbd_workspace_destroy(ws) == 0


# ---- batched `simulate` (Munthe-Kaas RK4 on the device) vs the numpy restatement of src/ode_integrators.jl:233-299 -----------
def quat_blocks(model):
    from rigidbodydynamics_jl_amd.mechanism import JOINT_QUAT_FLOATING, JOINT_QUAT_SPHERICAL
    return [int(model.q_offset[i]) for i in range(model.n_bodies) if int(model.joint_type[i]) in (JOINT_QUAT_FLOATING, JOINT_QUAT_SPHERICAL)]


def canon_q(model, q):
    """quaternions are compared up to sign (q and -q are the same rotation)."""
    q = q.copy()
    for o in quat_blocks(model):
        s = np.sign(q[:, o:o + 1])
        s[s == 0] = 1
        q[:, o:o + 4] *= s
    return q


@pytest.mark.parametrize("name", ["atlas_floating", "double_pendulum", "acrobot_urdf", "randmech1", "inner_floating"])
def test_simulate_matches_oracle_f64(rbd, oracle, models, name):
    """(Round 6: left to itself `simulate` of a mechanism the walk kernels take runs on the walk program compiled for it with the four stages of a step in ONE
    launch at every batch — measured ahead of the banked kernel with the stage fused in from 256 states up: Atlas fp64 4096 states 104.6 against 109.9 us per
    step, fp32 8192: 84.6 against 120.7 — so this small batch goes there when the program is in the cache.)"""
    import simulate_np
    model = models[name]
    B, dt, T = 6, 1e-3, 0.0095
    q, v, tau, _ = rand_inputs(rbd, model, B, 51, fext=True)
    state = rbd.MechanismState(model, B)
    rbd.set_configuration_(state, q)
    rbd.set_velocity_(state, v)
    ts = rbd.simulate_(state, T, dt=dt, torques=dev(tau, state))
    if name == "atlas_floating" and rbd.jit_precompile(models["double_pendulum"], torch.float32)[0]:
        assert "four stages per launch" in rbd.last_kernel(state), rbd.last_kernel(state)
    ts_ref, q_ref, v_ref = simulate_np.simulate(model, q, v, T, dt, tau)
    assert len(ts) == len(ts_ref) == 11
    qg, vg = host(state.q, state), host(state.v, state)
    assert np.abs(canon_q(model, qg) - canon_q(model, q_ref)).max() <= 1e-11 * max(1.0, np.abs(q_ref).max())
    assert np.abs(vg - v_ref).max() <= 1e-9 * max(1.0, np.abs(v_ref).max())


@pytest.mark.parametrize("batch", [6, 300])
@pytest.mark.parametrize("name", ["inner_floating", "mixed20", "randmech1"])
def test_simulate_fp32_with_spherical_joints(rbd, oracle, models, name, batch):
    """fp32 and QuaternionSpherical joints: the Bortz factor of the local-coordinate rates (spatial/util.jl:88-102) divides by 1 - cos θ, which is exactly zero in
    fp32 for θ < 3e-4 — and θ is rounding noise whenever q = q0, i.e. at the first stage of EVERY step (NaN for one state in ten before the small-angle series,
    rbd_integrator.hpp).  The lane-per-body kernels with the stage folded in (6 states) and the stage kernels (300), fp32 against the fp64 oracle."""
    import simulate_np
    model = models[name]
    B, dt, T = batch, 1e-3, 0.0045
    q, v, tau, _ = rand_inputs(rbd, model, B, 53, fext=True)
    state = rbd.MechanismState(model, B, dtype=torch.float32)
    rbd.set_configuration_(state, q)
    rbd.set_velocity_(state, v)
    rbd.simulate_(state, T, dt=dt, torques=dev(tau, state))
    qg, vg = host(state.q, state), host(state.v, state)
    assert np.isfinite(qg).all() and np.isfinite(vg).all()
    sel = np.r_[0:3, B - 3:B]
    _, q_ref, v_ref = simulate_np.simulate(model, q[sel].astype(np.float32).astype(np.float64), v[sel].astype(np.float32).astype(np.float64), T, dt, tau[sel].astype(np.float32).astype(np.float64))
    assert np.abs(canon_q(model, qg[sel]) - canon_q(model, q_ref)).max() <= 2e-5 * max(1.0, np.abs(q_ref).max())
    assert np.abs(vg[sel] - v_ref).max() <= 2e-3 * max(1.0, np.abs(v_ref).max())


def test_simulate_with_controller_and_store(rbd, oracle, models):
    """control!(torques, t, state) is evaluated before every stage's dynamics! (src/simulate.jl:42-48)."""
    import simulate_np
    model = models["double_pendulum"]
    B, dt = 4, 1e-2
    q, v, tau = rand_inputs(rbd, model, B, 52)
    state = rbd.MechanismState(model, B)
    rbd.set_configuration_(state, q)
    rbd.set_velocity_(state, v)
    d_tau = dev(tau, state)
    calls = []

    def control_(torques, t, st):
        calls.append(t)
        torques.copy_(d_tau)   # constant controller, so the constant-torque oracle applies

    ts, qs, vs = rbd.simulate_(state, 0.05, control_=control_, dt=dt, store=True)
    _, q_ref, v_ref = simulate_np.simulate(model, q, v, 0.05, dt, tau)
    assert len(qs) == len(ts) and len(calls) == 4 * (len(ts) - 1)
    assert np.allclose(calls[:4], [0.0, 0.005, 0.005, 0.01])
    assert np.abs(host(state.q, state) - q_ref).max() <= 1e-11 and np.abs(host(state.v, state) - v_ref).max() <= 1e-10
    assert torch.equal(qs[-1], state.q)


@pytest.mark.parametrize("path", ["fused", "unfused"])
def test_simulate_device_side_controllers(rbd, oracle, models, path, monkeypatch):
    """simulate(state, T, control!) without the per-stage host round trip (round-2 review): an open-loop τ(t) table sampled at the four stage
    times control! is called with (src/simulate.jl:42-48, ode_integrators.jl:48-55) and a PD law evaluated on the stage state inside the
    dynamics launch — against the numpy Munthe-Kaas restatement driving the same controllers.  `fused`: the stage rides in the lane-per-body
    ABA launches; `unfused`: the large-batch arrangement (stage kernel, element-wise PD kernel, walk kernel), forced at a small batch."""
    import simulate_np
    if path == "unfused":
        tune(monkeypatch, walk_min_batch="1")  # read when the workspace is created
    else:
        tune(monkeypatch, sim_walk_min_batch=1 << 40)  # (round 6: left to itself `simulate` takes the looped walk program at every batch; this test is about the lane-per-body kernels with the stage fused in)
    model = models["atlas_fixed"]
    B, dt, T = 5, 2e-3, 0.0075
    nsteps = 4
    q, v, _ = rand_inputs(rbd, model, B, 57)
    rng = np.random.default_rng(58)
    # ---- table, per stage: τ(t) = A sin(ω t + φ) per dof and state
    A, om, ph = rng.random((B, model.nv)), 40 * rng.random((B, model.nv)), rng.random((B, model.nv))
    tau_of = lambda b, t: A[b] * np.sin(om[b] * t + ph[b])
    stage_t = np.array([k * dt + c * dt for k in range(nsteps) for c in (0.0, 0.5, 0.5, 1.0)])
    table = np.stack([A * np.sin(om * t + ph) for t in stage_t])  # (4 nsteps, B, nv)
    state = rbd.MechanismState(model, B)
    rbd.set_configuration_(state, q); rbd.set_velocity_(state, v)
    rbd.simulate_(state, T, control_=rbd.TorqueTable(torch.as_tensor(table).cuda(), per_stage=True), dt=dt)
    _, q_ref, v_ref = simulate_np.simulate(model, q, v, T, dt, control=lambda b, t, qq, vv: tau_of(b, t))
    assert np.abs(host(state.q, state) - q_ref).max() <= 1e-10 and np.abs(host(state.v, state) - v_ref).max() <= 1e-9
    # zero-order hold: one entry per step
    rbd.set_configuration_(state, q); rbd.set_velocity_(state, v)
    rbd.simulate_(state, T, control_=rbd.TorqueTable(torch.as_tensor(table[::4].copy()).cuda(), per_stage=False), dt=dt)
    # (the hold's reference: the step index is known by construction, so integrate step by step with constant torques)
    qh, vh = q.copy(), v.copy()
    for k in range(nsteps):
        _, qh, vh = simulate_np.simulate(model, qh, vh, dt / 2, dt, table[4 * k])
    assert np.abs(host(state.q, state) - qh).max() <= 1e-10 and np.abs(host(state.v, state) - vh).max() <= 1e-9
    # ---- PD law with a feed-forward term
    kp, kd = 2 * rng.random(model.nv), 0.02 * rng.random(model.nv)  # (the wrist links have 4e-4 kg m^2 about their axes: stiff gains would need a smaller step)
    qdes, tff = 0.3 * rng.standard_normal((B, model.nq)), rng.random((B, model.nv))
    rbd.set_configuration_(state, q); rbd.set_velocity_(state, v)
    rbd.simulate_(state, T, control_=rbd.PDControl(torch.as_tensor(kp), torch.as_tensor(kd), torch.as_tensor(qdes).cuda(), torch.as_tensor(tff).cuda()), dt=dt)
    _, q_ref, v_ref = simulate_np.simulate(model, q, v, T, dt, control=lambda b, t, qq, vv: tff[b] - kp * (qq - qdes[b]) - kd * vv)
    assert np.abs(host(state.q, state) - q_ref).max() <= 1e-10 and np.abs(host(state.v, state) - v_ref).max() <= 1e-9
    k = rbd.last_kernel(state)
    assert ("walk" in k) == (path == "unfused"), k


@pytest.mark.parametrize("layout", ["aos", "soa"])
@pytest.mark.parametrize("name", ["mixed20", "inner_floating", "randmech1"])
@pytest.mark.parametrize("dtype", ["f32", "f64", "f64_stash"])
def test_simulate_stage_folded_in_on_every_joint_type(rbd, oracle, models, name, layout, dtype, monkeypatch):
    """The same for mechanisms with Planar / QuaternionSpherical joints and QuaternionFloating joints below the world: only the lane-per-state kernel compiled for the
    mechanism takes them (its stage is generic over the joint types: the Bortz equation for the spherical joints, the SE(3) log / exp for every 6-dof joint) — in fp32
    and, since round 6, in fp64 (aba_spec_f64: the program in doubles of exactly these mechanisms; f64_stash: its sibling with the spare rows in the workspace's HBM
    stash, where the next q leaves from the lanes instead of through spare rows)."""
    import simulate_np
    stash = dtype == "f64_stash"
    dtype = dtype[:3]
    tune(monkeypatch, spec_aba_min_batch=1, spec_f64_stash=1 if stash else 0)
    model = models[name]
    B, dt, nsteps = 70, 1e-3, 3
    T = (nsteps - 0.5) * dt
    state, q, v, tau, _ = make(rbd, model, B, dtype, layout, 93, fext=False)
    sel = np.r_[0:3, B - 2:B]
    try:
        rbd.simulate_(state, T, dt=dt, torques=dev(tau, state))
    except rbd._capi.RBDError as e:
        if e.status == 3:
            pytest.skip("hiprtc not available")
        raise
    k = rbd.last_kernel(state)
    if "folded in" not in k:
        pytest.skip("no compiled kernel for this route on this box: " + k)
    assert ("aba_spec_gst_f64" if stash else "aba_spec_" + dtype) in k, k
    ref = simulate_np.simulate(model, q[sel], v[sel], T, dt, tau[sel])
    qg, vg = host(state.q, state)[sel], host(state.v, state)[sel]
    assert np.isfinite(host(state.q, state)).all() and np.isfinite(host(state.v, state)).all()
    tq, tv = (2e-5, 2e-3) if dtype == "f32" else (1e-10, 1e-9)
    assert np.abs(canon_q(model, qg) - canon_q(model, ref[1])).max() <= tq * max(1.0, np.abs(ref[1]).max())
    assert np.abs(vg - ref[2]).max() <= tv * max(1.0, np.abs(ref[2]).max())


ONE_LAUNCH_LANE_PER_STATE = False  # (aba_spec_f32: four launches per step)


@pytest.mark.parametrize("layout", ["aos", "soa"])
@pytest.mark.parametrize("kernel,name", [("walk_spec_f64", "atlas_floating"), ("walk_spec_f32", "atlas_floating"), ("walk_spec_f32x2", "atlas_floating"), ("aba_spec_f32", "atlas_floating"),
                                         ("walk_spec_f32", "limbs_humanoid"), ("walk_spec_f32x2", "limbs_humanoid"), ("walk_spec_f64", "limbs_only_children")])
@pytest.mark.parametrize("launches", ["one_launch_per_step", "four_launches_per_step"])
def test_simulate_stage_folded_into_the_compiled_kernels(rbd, oracle, models, kernel, name, layout, launches, monkeypatch):
    """Large batches: the Munthe-Kaas stage of `simulate` (src/ode_integrators.jl:233-299) inside the dynamics! kernels compiled for the mechanism
    (csrc/rbd_mk_fuse.hpp: four launches per step, no stage kernels) — forced at a small ragged batch so that states can be compared with the numpy
    restatement of the integrator: Atlas with its floating base (the SE(3) log / exp path), constant torques, the torque table at the stage times, the PD law
    on the stage state; fp64 at 1e-10, fp32 against the fp64 oracle.  limbs_humanoid / limbs_only_children: the same through prismatic, sin-cos and fixed joints
    (the sin-cos joint's two coordinates per velocity in the stage's arithmetic; the PD law leaves sin-cos joints alone)."""
    import simulate_np
    dtype = "f64" if kernel.endswith("f64") else "f32"
    knobs = dict(walk_min_batch=1, spec_walk_min_batch=1, walk_pair_min_batch=1 if kernel == "walk_spec_f32x2" else 1 << 40,
                 spec_aba_min_batch=1 if kernel == "aba_spec_f32" else 1 << 40)
    # round 6: the walk kernels' four-stages-per-launch program (LOOP = true) is admitted statically (csrc/rbd_jit.hip jit_walk_admit) and no longer compared with
    # the single-stage kernel inside rbd_simulate; both forms are held against the oracle here — constant torques, the torque table (the per-stage reload between
    # the passes) and the PD law (evaluated between the passes), which the round-5 run-time check never exercised
    one_launch = launches == "one_launch_per_step"
    if one_launch and kernel == "aba_spec_f32" and not ONE_LAUNCH_LANE_PER_STATE:
        pytest.skip("the lane-per-state kernel takes four launches per step")
    tune(monkeypatch, sim_one_launch=1 if one_launch else 0, **knobs)
    model = models[name]
    B, dt, nsteps = 70, 1e-3, 3
    T = (nsteps - 0.5) * dt
    state, q, v, tau, _ = make(rbd, model, B, dtype, layout, 91, fext=False)
    sel = np.r_[0:3, B - 2:B]  # (the oracle integrates one state at a time)
    tq, tv = (1e-10, 1e-9) if dtype == "f64" else (2e-5, 2e-3)

    def check(ref, what):
        assert "Munthe-Kaas stage folded in" in rbd.last_kernel(state) and ("aba_spec_f32" in rbd.last_kernel(state)) == (kernel == "aba_spec_f32"), rbd.last_kernel(state)
        assert ("four stages per launch" in rbd.last_kernel(state)) == one_launch, rbd.last_kernel(state)
        qg, vg = host(state.q, state)[sel], host(state.v, state)[sel]
        assert np.isfinite(host(state.q, state)).all() and np.isfinite(host(state.v, state)).all(), what
        assert np.abs(canon_q(model, qg) - canon_q(model, ref[1])).max() <= tq * max(1.0, np.abs(ref[1]).max()), what
        assert np.abs(vg - ref[2]).max() <= tv * max(1.0, np.abs(ref[2]).max()), what

    def reset():
        rbd.set_configuration_(state, q); rbd.set_velocity_(state, v)

    try:
        rbd.simulate_(state, T, dt=dt, torques=dev(tau, state))
    except rbd._capi.RBDError as e:
        if e.status == 3:
            pytest.skip("hiprtc not available")
        raise
    if "folded in" not in rbd.last_kernel(state):
        pytest.skip("no compiled kernel for this route on this box: " + rbd.last_kernel(state))
    check(simulate_np.simulate(model, q[sel], v[sel], T, dt, tau[sel]), "constant torques")
    # the torque table sampled at the four stage times
    rng = np.random.default_rng(92)
    A, om, ph = rng.random((B, model.nv)), 40 * rng.random((B, model.nv)), rng.random((B, model.nv))
    stage_t = np.array([k * dt + c * dt for k in range(nsteps) for c in (0.0, 0.5, 0.5, 1.0)])
    table = np.stack([A * np.sin(om * t + ph) for t in stage_t]).astype(ND[dtype])  # (4 nsteps, B, nv)
    tab_dev = torch.as_tensor(table if layout == "aos" else np.ascontiguousarray(table.transpose(0, 2, 1))).cuda()
    reset()
    rbd.simulate_(state, T, control_=rbd.TorqueTable(tab_dev, per_stage=True), dt=dt)
    tau_of = lambda b, t: (A[sel][b] * np.sin(om[sel][b] * t + ph[sel][b])).astype(ND[dtype]).astype(np.float64)
    check(simulate_np.simulate(model, q[sel], v[sel], T, dt, control=lambda b, t, qq, vv: tau_of(b, t)), "torque table")
    # the PD law on the stage state, with a feed-forward term (revolute joints; the floating base takes the feed-forward term alone)
    kp, kd = (2 * rng.random(model.nv)).astype(ND[dtype]).astype(np.float64), (0.02 * rng.random(model.nv)).astype(ND[dtype]).astype(np.float64)
    qdes = (0.3 * rng.standard_normal((B, model.nq))).astype(ND[dtype]).astype(np.float64)
    tff = rng.random((B, model.nv)).astype(ND[dtype]).astype(np.float64)
    reset()
    rbd.simulate_(state, T, control_=rbd.PDControl(torch.as_tensor(kp), torch.as_tensor(kd), dev(qdes, state), dev(tff, state)), dt=dt)
    one = np.array([1.0 if t in (1, 2) else 0.0 for t in model.joint_type for _ in range({0: 0, 3: 6}.get(int(t), 1))])  # revolute / prismatic coordinates (v indexing)
    qidx = np.array([int(model.q_offset[b]) for b in range(model.n_bodies) for _ in range({0: 0, 3: 6}.get(int(model.joint_type[b]), 1))])  # q coordinate of a 1-dof v coordinate

    def pd(b, t, qq, vv):
        return tff[sel][b] - one * (kp * (qq[qidx] - qdes[sel][b][qidx]) + kd * vv)
    check(simulate_np.simulate(model, q[sel], v[sel], T, dt, control=pd), "PD law")


def test_simulate_four_bar_loops(rbd, oracle, models):
    """The loop-joint branch inside the integrator: closure is kept (no stabilization) like test/test_simulate.jl:203-213, here
    for 0.1 s on the GPU, against the oracle's integration of the same states."""
    import simulate_np
    model = models["four_bar"]
    B, dt = 8, 1e-3
    q = np.tile(rbd.FOUR_BAR_INITIAL_Q, (B, 1))
    v = np.zeros((B, 3))
    state = rbd.MechanismState(model, B)
    rbd.set_configuration_(state, q)
    rbd.set_velocity_(state, v)
    rbd.simulate_(state, 0.0995, dt=dt, stabilization_gains=None)
    _, q_ref, v_ref = simulate_np.simulate(model, q[:1], v[:1], 0.0995, dt, stabilize=False)
    assert np.abs(host(state.q, state) - q_ref).max() <= 1e-10 and np.abs(host(state.v, state) - v_ref).max() <= 1e-9


def test_simulate_energy_full_batch(rbd, oracle, models):
    """Size-independent property at B = 4096: free motion conserves KE + PE (checked with the oracle's energy on a sample)."""
    model = models["atlas_floating"]
    B = 4096
    q, v, _ = rand_inputs(rbd, model, B, 53)
    state = rbd.MechanismState(model, B)
    rbd.set_configuration_(state, q)
    rbd.set_velocity_(state, v)
    rbd.simulate_(state, 0.0195, dt=1e-3)
    n = 64
    ke0, pe0 = oracle.energy(model, q[:n], v[:n])
    ke1, pe1 = oracle.energy(model, host(state.q, state)[:n], host(state.v, state)[:n])
    assert (np.abs(ke1 + pe1 - ke0 - pe0) / (np.abs(ke0) + np.abs(pe0))).max() < 1e-7
    qn = state.q[:, :4].norm(dim=1)
    assert float((qn - 1).abs().max()) < 1e-12


def test_dynamics_ode_form(rbd, oracle, models):
    """dynamics!(ẋ, result, state, x, …) — src/mechanism_algorithms.jl:880-889 / test/test_mechanism_algorithms.jl:755-771."""
    model = models["atlas_floating"]
    B = 9
    q, v, tau = rand_inputs(rbd, model, B, 61)
    state = rbd.MechanismState(model, B)
    result = rbd.DynamicsResult(model, B)
    x = torch.as_tensor(np.hstack([q, v])).cuda()
    xd = torch.zeros_like(x)
    rbd.dynamics_ode_(xd, result, state, x, dev(tau, state))
    vd, qd = oracle.dynamics(model, q, v, tau, want_qdot=True)
    ref = np.hstack([qd, vd])
    assert np.abs(xd.cpu().numpy() - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("name", ["atlas_floating", "randmech2", "inner_floating"])
def test_mass_matrix_solve_aba_equals_cholesky(rbd, oracle, models, name):
    """x = M⁻¹ rhs through the O(n) articulated-body solve and through CRBA + Cholesky agree (fp64, 1e-10)."""
    model = models[name]
    B = 41
    state, q, v, rhs, _ = make(rbd, model, B, "f64", "aos", 71)
    xa, xc = torch.zeros_like(state.v), torch.zeros_like(state.v)
    rbd.mass_matrix_solve_(xa, state, dev(rhs, state), algorithm="aba")
    rbd.mass_matrix_solve_(xc, state, dev(rhs, state), algorithm="cholesky")
    M = oracle.mass_matrix(model, q)
    Ms = np.tril(M) + np.transpose(np.tril(M, -1), (0, 2, 1))
    xr = np.linalg.solve(Ms, rhs[..., None])[..., 0]
    for x in (xa, xc):
        assert np.abs(host(x, state) - xr).max() <= 1e-9 * max(1.0, np.abs(xr).max())


@pytest.mark.parametrize("name", ["double_pendulum", "acrobot_urdf", "atlas_fixed", "atlas_floating", "randmech3"])
def test_cholesky_route_f32_mfma(rbd, oracle, models, name):
    """fp32 CRBA + Cholesky route: for nv <= 40 the factorization runs on the matrix cores (chol_mfma_kernel, tile sizes
    NT = 1, 1, 8, 9, 10 here).  Checked by the backward error of M x = tau - c with the fp64 oracle's M and c."""
    model = models[name]
    B = 50  # not a multiple of the 16 states a wavefront factors at once
    state, q, v, tau, fe = make(rbd, model, B, "f32", "aos", 81)
    result = rbd.DynamicsResult(model, B, dtype=torch.float32)
    rbd.dynamics_(result, state, dev(tau, state), dev(fe, state), algorithm="crba")
    assert rbd.sync(state) == 0
    xg = host(result.vd, state)
    M = oracle.mass_matrix(model, q)
    Ms = np.tril(M) + np.transpose(np.tril(M, -1), (0, 2, 1))
    rhs = tau - oracle.dynamics_bias(model, q, v, fe)
    res = np.einsum("bij,bj->bi", Ms, xg) - rhs
    eta = np.linalg.norm(res, axis=1) / (np.linalg.norm(Ms, axis=(1, 2)) * np.linalg.norm(xg, axis=1) + np.linalg.norm(rhs, axis=1))
    assert np.isfinite(xg).all() and eta.max() <= 2e-5, eta.max()


@pytest.mark.parametrize("dtype,name", [("f32", "atlas_floating"), ("f32", "atlas_fixed"), ("f32", "randmech1"), ("f64", "atlas_floating"),
                                        ("f64", "double_pendulum")])
def test_cholesky_solve_factor_and_solution(rbd, models, dtype, name):
    """rbd_cholesky_solve on caller-provided SPD matrices: L equals LAPACK's potrf factor, x = A⁻¹ b (fp32, nv <= 40: the MFMA
    tile kernel; fp64: the register-resident kernel)."""
    import ctypes
    from rigidbodydynamics_jl_amd import _capi
    model = models[name]
    nv, B = model.nv, 37
    rng = np.random.default_rng(91)
    A = rng.standard_normal((B, nv, nv))
    A = A @ A.transpose(0, 2, 1) + nv * np.eye(nv)
    b = rng.standard_normal((B, nv))
    st = rbd.MechanismState(model, B, dtype=TD[dtype])
    lower_colmajor = np.tril(A).transpose(0, 2, 1).reshape(B, nv * nv)   # only the lower triangle is provided
    M = torch.as_tensor(lower_colmajor.copy(), dtype=TD[dtype]).cuda()
    rhs = torch.as_tensor(b, dtype=TD[dtype]).cuda()
    x, L = torch.zeros_like(rhs), torch.zeros_like(M)
    opts = st._opts()
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    assert _capi.lib().rbd_cholesky_solve(st.ws.handle, B, P(M), P(rhs), P(x), P(L), ctypes.byref(opts)) == 0
    assert rbd.sync(st) == 0
    Lg = np.tril(L.double().cpu().numpy().reshape(B, nv, nv).transpose(0, 2, 1))
    tol = 1e-12 if dtype == "f64" else 2e-5
    assert np.abs(Lg - np.linalg.cholesky(A)).max() <= tol * np.abs(A).max() ** 0.5
    xr = np.linalg.solve(A, b[..., None])[..., 0]
    assert np.abs(x.double().cpu().numpy() - xr).max() <= tol * max(1.0, np.abs(xr).max())


@pytest.mark.parametrize("dtype,n_rev", [("f64", 44), ("f64", 58), ("f32", 50), ("f32", 58)])
def test_cholesky_route_lds_kernel_nv_49_to_64(rbd, oracle, dtype, n_rev):
    """Systems with 49..64 velocities take chol_solve_kernel (factor resident in LDS, one state per wavefront); in fp64 its four states per
    workgroup need up to 141 KB of LDS, above the default 64 KB dynamic limit, so the launcher must raise the kernel's limit first."""
    rng = np.random.default_rng(31 + n_rev)
    mech = rbd.rand_tree_mechanism(rng, ["QuaternionFloating"] + ["Revolute"] * n_rev, lambda m, r: m.bodies[max(1, len(m.bodies) - 1 - int(r.integers(0, 3)))])  # at most 3 children per body
    model = rbd.flatten(mech)
    assert 49 <= model.nv <= 64
    B = 21
    state, q, v, tau, fe = make(rbd, model, B, dtype, "aos", 77)
    result = rbd.DynamicsResult(model, B, dtype=TD[dtype])
    rbd.dynamics_(result, state, dev(tau, state), dev(fe, state), algorithm="crba")
    ref = oracle.dynamics(model, q, v, tau, fe)
    got = host(result.vd, state)
    if dtype == "f64":
        assert np.abs(got - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max())
    else:  # fp32: backward error of the solve against the oracle's M and bias
        Mr, cr = oracle.mass_matrix(model, q), oracle.dynamics_bias(model, q, v, fe)
        Mr = np.tril(Mr) + np.transpose(np.tril(Mr, -1), (0, 2, 1))  # the oracle fills the lower triangle, like the reference
        r = np.einsum("bij,bj->bi", Mr, got) - (tau - cr)
        denom = np.linalg.norm(Mr, axis=(1, 2)) * np.linalg.norm(got, axis=1) + np.linalg.norm(tau - cr, axis=1)
        assert (np.linalg.norm(r, axis=1) / denom).max() <= 2e-5


def test_config4_shard_f32_full_size_round_trip(rbd, oracle, models):
    """BASELINE configs[3]: one GPU's shard of the 524 288-state fp32 batch (65 536 states).  Size-independent property on the
    GPU (dynamics! then inverse_dynamics! returns τ, in fp32 backward-error terms) plus an oracle check on a sample."""
    model = models["atlas_floating"]
    B = 65536
    state, q, v, tau, _ = make(rbd, model, B, "f32", "aos", 3)
    result = rbd.DynamicsResult(model, B, dtype=torch.float32)
    t = dev(tau, state)
    rbd.dynamics_(result, state, t)
    back = torch.zeros_like(state.v)
    rbd.inverse_dynamics_(back, state, result.vd)
    c = torch.zeros_like(state.v)
    rbd.dynamics_bias_(c, state)
    rel = (back - t).norm(dim=1) / (t - c).norm(dim=1)
    assert float(rel.max()) <= 5e-4          # fp32 RNEA(ABA(τ)) − τ, cond(M) ≈ 5e5
    vd = host(result.vd, state)  # the whole shard against the oracle: fp32 backward error ||M v̇ + c − τ|| / ||τ − c|| with the fp64 oracle's RNEA
    resid = oracle.inverse_dynamics(model, q, v, vd, nthreads=NT) - tau
    cc = oracle.dynamics_bias(model, q, v, nthreads=NT)
    assert (np.linalg.norm(resid, axis=1) / np.linalg.norm(tau - cc, axis=1)).max() <= 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["aos", "soa"])
@pytest.mark.parametrize("name", MODELS)
def test_kinematics_byproducts_f64(rbd, oracle, models, name, layout):
    """momentum_matrix!, center_of_mass, kinetic_energy, gravitational_potential_energy
    (test/test_mechanism_algorithms.jl:527-545, :564-572; tolerance 1e-12 relative as in the reference test)."""
    model = models[name]
    B = 45
    state, q, v, _, _ = make(rbd, model, B, "f64", layout, 31)
    A = torch.zeros((B, 6 * model.nv) if layout == "aos" else (6 * model.nv, B), dtype=torch.float64, device="cuda")
    rbd.momentum_matrix_(A, state)
    com = rbd.center_of_mass(state)
    ke, pe = rbd.kinetic_energy(state), rbd.gravitational_potential_energy(state)
    torch.cuda.synchronize()
    A_ref, _, com_ref = oracle.momentum_matrix(model, q, v)
    ke_ref, pe_ref = oracle.energy(model, q, v)
    got = host(A, state).reshape(B, model.nv, 6).transpose(0, 2, 1)
    assert np.abs(got - A_ref).max() <= 1e-12 * max(1.0, np.abs(A_ref).max())
    assert np.abs(host(com, state) - com_ref).max() <= 1e-12 * max(1.0, np.abs(com_ref).max())
    assert np.abs(ke.cpu().numpy() - ke_ref).max() <= 1e-12 * max(1.0, np.abs(ke_ref).max())
    assert np.abs(pe.cpu().numpy() - pe_ref).max() <= 1e-12 * max(1.0, np.abs(pe_ref).max())


@pytest.mark.gpu
def test_kinematics_byproducts_f32_and_full_size(rbd, oracle, models):
    model = models["atlas_floating"]
    state, q, v, _, _ = make(rbd, model, 64, "f32", "aos", 32)
    A = torch.zeros((64, 6 * model.nv), dtype=torch.float32, device="cuda")
    rbd.momentum_matrix_(A, state)
    A_ref, _, com_ref = oracle.momentum_matrix(model, q, v)
    got = host(A, state).reshape(64, model.nv, 6).transpose(0, 2, 1)
    assert np.abs(got - A_ref).max() <= 2e-5 * np.abs(A_ref).max()
    assert np.abs(host(rbd.center_of_mass(state), state) - com_ref).max() <= 1e-5
    # full size, CPU-free: A v (momentum) against the floating-base rows of M v — for a floating root joint the first six
    # rows of M v are the total momentum expressed in the base frame, so |h| must agree; and KE = v'(Mv)/2.
    B = 4096
    state, q, v, _, _ = make(rbd, model, B, "f64", "aos", 33)
    A = torch.zeros((B, 6 * model.nv), dtype=torch.float64, device="cuda")
    rbd.momentum_matrix_(A, state)
    result = rbd.DynamicsResult(model, B)
    rbd.mass_matrix_(result, state)
    Mv = torch.einsum("bij,bj->bi", result.massmatrix_dense(), state.v)
    ke = rbd.kinetic_energy(state)
    assert float((0.5 * (Mv * state.v).sum(1) - ke).abs().max()) <= 1e-10 * float(ke.abs().max())
    h = torch.einsum("bki,bi->bk", A.view(B, model.nv, 6).transpose(1, 2), state.v)
    # rotate the angular/linear parts back into the base frame: only norms of the force part are frame independent
    assert float((h[:, 3:].norm(dim=1) - Mv[:, 3:6].norm(dim=1)).abs().max()) <= 1e-9 * float(h[:, 3:].norm(dim=1).max())


# ---- chain-scheduled ABA (RBD_ALGO_ABA_CHAINS): the same dynamics! through the other lane mapping -------------------------
CHAIN_MODELS = ["atlas_floating", "atlas_fixed", "valkyrie_floating", "double_pendulum", "acrobot_urdf"]
# the lane mappings of the fused ABA beside one body per lane (the chain / track / pipe mappings of rounds 1-2 lost at every batch size and were removed:
# DESIGN.md §8; their RBD_ALGO_* values stay reserved and return RBD_ERR_UNSUPPORTED)
MAPPINGS = ["aba_walk", "aba_banks"]




# the banked and walk kernels exist twice: built with the library (they interpret the level structure / the plan; RBD_JIT=0 here) and compiled for the mechanism at run time
JIT = ["compiled", "built"]


@pytest.mark.gpu
@pytest.mark.parametrize("jit", JIT)
@pytest.mark.parametrize("algorithm", MAPPINGS)
@pytest.mark.parametrize("layout", ["aos", "soa"])
@pytest.mark.parametrize("name", CHAIN_MODELS)
def test_dynamics_chains_f64(rbd, oracle, models, name, layout, algorithm, jit, monkeypatch):
    monkeypatch.setenv("RBD_JIT", "1" if jit == "compiled" else "0")
    model = models[name]
    B = 67  # ragged against every states-per-wave (64, 32, 16, 4)
    state, q, v, tau, fe = make(rbd, model, B, "f64", layout, 41)
    result = rbd.DynamicsResult(model, B, layout=layout)
    rbd.dynamics_(result, state, dev(tau, state), dev(fe, state), algorithm=algorithm)
    torch.cuda.synchronize()
    ref, qd_ref = oracle.dynamics(model, q, v, tau, fe, want_qdot=True)
    got = host(result.vd, state)
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max())
    assert np.abs(host(result.qd, state) - qd_ref).max() <= 1e-13 * max(1.0, np.abs(qd_ref).max())
    # and against the lane-per-body mapping, no torques / wrenches (defaults)
    r1, r2 = rbd.DynamicsResult(model, B, layout=layout), rbd.DynamicsResult(model, B, layout=layout)
    rbd.dynamics_(r1, state, algorithm=algorithm)
    rbd.dynamics_(r2, state, algorithm="aba_lanes")
    assert float((r1.vd - r2.vd).abs().max()) <= 1e-10 * max(1.0, float(r2.vd.abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("algorithm", MAPPINGS)
@pytest.mark.parametrize("B", [1, 3, 4, 5, 15, 16, 17, 1000])
def test_dynamics_chains_batch_sizes_and_f32(rbd, oracle, models, B, algorithm):
    model = models["atlas_floating"]
    state, q, v, tau, fe = make(rbd, model, B, "f64", "aos", 42 + B)
    result = rbd.DynamicsResult(model, B)
    rbd.dynamics_(result, state, dev(tau, state), dev(fe, state), algorithm=algorithm)
    ref = oracle.dynamics(model, q, v, tau, fe)
    assert np.abs(host(result.vd, state) - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max())
    state, q, v, tau, fe = make(rbd, model, B, "f32", "aos", 43 + B)
    result = rbd.DynamicsResult(model, B, dtype=torch.float32)
    rbd.dynamics_(result, state, dev(tau, state), dev(fe, state), algorithm=algorithm)
    ref = oracle.dynamics(model, q, v, tau, fe)
    got = host(result.vd, state)
    assert np.isfinite(got).all()
    assert_fp32_forward(oracle, model, q, got, ref, what=algorithm)  # same criterion as test_dynamics_f32


@pytest.mark.gpu
@pytest.mark.parametrize("algorithm", MAPPINGS)
def test_dynamics_chains_random_trees(rbd, oracle, algorithm):
    """Random revolute / prismatic / fixed / sin-cos trees (with and without a floating root), as the reference's randomized tests do."""
    from test_chain_plan import random_tree
    rng = np.random.default_rng(9)
    for trial in range(12):
        mech = random_tree(rbd, rng, int(rng.integers(1, 30)), bool(trial % 2), float(rng.uniform(0, 1)))
        model = rbd.flatten(mech)
        B = 33
        state, q, v, tau, fe = make(rbd, model, B, "f64", "aos", 50 + trial)
        result = rbd.DynamicsResult(model, B)
        try:
            rbd.dynamics_(result, state, dev(tau, state), dev(fe, state), algorithm=algorithm)
        except Exception:
            # banks: a tree too shallow or too small to split into two banks that save lanes; tracks: a chain so long that its
            # per-step LDS rows exceed one CU's 160 KB (RBD_ERR_UNSUPPORTED, the default then takes another mapping)
            assert algorithm in ("aba_banks", "aba_walk")  # walk: more than 11 steps per track
            continue
        ref = oracle.dynamics(model, q, v, tau, fe)
        got = host(result.vd, state)
        assert np.isfinite(got).all(), trial
        assert np.abs(got - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max()), trial


@pytest.mark.gpu
def test_dynamics_chains_scope_and_auto_selection(rbd, oracle, models):
    model = models["randmech1"]  # has 3-dof joints: outside the walk mapping
    state, *_ = make(rbd, model, 8, "f64", "aos", 60)
    result = rbd.DynamicsResult(model, 8)
    with pytest.raises(Exception):
        rbd.dynamics_(result, state, algorithm="aba_walk")
    with pytest.raises(Exception):
        rbd.dynamics_(result, state, algorithm="aba_chains")  # the removed round-1 mapping: RBD_ERR_UNSUPPORTED
    rbd.dynamics_(result, state)  # default still works (lanes)
    # full size: both mappings agree with each other and the dynamics! -> inverse_dynamics round trip closes
    # (test/test_mechanism_algorithms.jl:729-740); at this size the default (RBD_ALGO_ABA) picks the banked mapping
    model = models["atlas_floating"]
    B = 32768
    state, q, v, tau, fe = make(rbd, model, B, "f64", "aos", 61)
    r0, r1 = rbd.DynamicsResult(model, B), rbd.DynamicsResult(model, B)
    t = dev(tau, state)
    rbd.dynamics_(r0, state, t, algorithm="aba_walk")
    rbd.dynamics_(r1, state, t, algorithm="aba_lanes")
    assert float((r0.vd - r1.vd).abs().max()) <= 1e-9 * float(r1.vd.abs().max())
    back = torch.zeros_like(t)
    rbd.inverse_dynamics_(back, state, r0.vd)
    assert float((back - t).abs().max()) <= 1e-8 * max(1.0, float(t.abs().max()))
    state, q, v, tau, fe = make(rbd, model, B, "f32", "aos", 62)
    r0, r1 = rbd.DynamicsResult(model, B, dtype=torch.float32), rbd.DynamicsResult(model, B, dtype=torch.float32)
    t = dev(tau, state)
    rbd.dynamics_(r0, state, t)
    rbd.dynamics_(r1, state, t, algorithm="aba_lanes")
    assert float((r0.vd - r1.vd).abs().max()) <= 2e-3 * float(r1.vd.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("jit", JIT)
@pytest.mark.parametrize("layout", ["aos", "soa"])
@pytest.mark.parametrize("name", MODELS)
def test_inverse_dynamics_and_bias_banked_f64(rbd, oracle, models, name, layout, jit, monkeypatch):
    monkeypatch.setenv("RBD_JIT", "1" if jit == "compiled" else "0")
    """The two-bodies-per-lane RNEA (every tree joint type) against the oracle and against the one-body-per-lane kernel."""
    model = models[name]
    B = 45
    state, q, v, tau, fe = make(rbd, model, B, "f64", layout, 71)
    rng = np.random.default_rng(72)
    vd = rng.standard_normal((B, model.nv))
    out = torch.zeros_like(dev(tau, state))
    try:
        rbd.inverse_dynamics_(out, state, dev(vd, state), dev(fe, state), mapping="banks")
    except Exception:
        pytest.skip("tree too small to split into two banks")
    ref = oracle.inverse_dynamics(model, q, v, vd, fe)
    assert np.abs(host(out, state) - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max())
    out2 = torch.zeros_like(out)
    rbd.dynamics_bias_(out, state, mapping="banks")
    rbd.dynamics_bias_(out2, state, mapping="lanes")
    ref = oracle.dynamics_bias(model, q, v)
    assert np.abs(host(out, state) - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max())
    assert float((out - out2).abs().max()) <= 1e-10 * max(1.0, float(out2.abs().max()))


@pytest.mark.gpu
def test_inverse_dynamics_banked_f32_and_full_size(rbd, oracle, models):
    model = models["atlas_floating"]
    state, q, v, tau, fe = make(rbd, model, 64, "f32", "aos", 73)
    vd = np.random.default_rng(74).standard_normal((64, model.nv)).astype(np.float32).astype(np.float64)
    out = torch.zeros_like(dev(tau, state))
    rbd.inverse_dynamics_(out, state, dev(vd, state), dev(fe, state), mapping="banks")
    ref = oracle.inverse_dynamics(model, q, v, vd, fe)
    assert np.abs(host(out, state) - ref).max() <= 2e-4 * max(1.0, np.abs(ref).max())
    B = 16384  # auto picks the walk kernels (banked below 8193 states); dynamics! -> inverse_dynamics closes (test/test_mechanism_algorithms.jl:729-740)
    state, q, v, tau, fe = make(rbd, model, B, "f64", "aos", 75)
    res = rbd.DynamicsResult(model, B)
    t = dev(tau, state)
    rbd.dynamics_(res, state, t)
    back = torch.zeros_like(t)
    rbd.inverse_dynamics_(back, state, res.vd)
    assert float((back - t).abs().max()) <= 1e-8 * max(1.0, float(t.abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["aos", "soa"])
@pytest.mark.parametrize("name", MODELS)
def test_geometric_jacobian_f64(rbd, oracle, models, name, layout):
    """geometric_jacobian!(out, state, path) in the root frame against the oracle (test/test_mechanism_algorithms.jl:310-327)."""
    model = models[name]
    B = 37
    state, q, v, _, _ = make(rbd, model, B, "f64", layout, 81)
    rng = np.random.default_rng(82)
    J = torch.full((B, 6 * model.nv) if layout == "aos" else (6 * model.nv, B), 7.0, dtype=torch.float64, device="cuda")  # off-path columns must be zeroed
    for _ in range(6):
        base, body = rng.choice(np.arange(-1, model.n_bodies), 2, replace=False)
        rbd.geometric_jacobian_(J, state, int(base), int(body))
        ref, trel = oracle.geometric_jacobian(model, q, base, body, v)
        got = host(J, state).reshape(B, model.nv, 6).transpose(0, 2, 1)
        assert np.abs(got - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max())
        assert np.abs(np.einsum("bkn,bn->bk", got, v) - trel).max() <= 1e-11 * max(1.0, np.abs(trel).max())
    with pytest.raises(ValueError):
        rbd.geometric_jacobian_(J, state, 0, model.n_bodies)


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["aos", "soa"])
@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("name", ["atlas_floating", "atlas_fixed", "valkyrie_floating", "randmech1", "inner_floating", "mixed20", "limbs_humanoid"])
def test_kinematics_byproducts_compiled_for_the_mechanism(rbd, oracle, models, name, dtype, layout, monkeypatch):
    """Round 6: momentum_matrix! / center_of_mass / energies, geometric_jacobian!, momentum / momentum_rate_bias one lane per state, compiled for the mechanism
    (csrc/rbd_spec.hpp kin_spec<T, WHAT>: what large batches take — forced at a small ragged one here) against the oracle, on every tree joint type, both layouts;
    fp64 at the reference's 1e-12 (test/test_mechanism_algorithms.jl:527-545), fp32 against the fp64 oracle."""
    tune(monkeypatch, spec_kin_min_batch=1)
    model = models[name]
    B = 70
    tol = 1e-12 if dtype == "f64" else 3e-4
    state, q, v, _, _ = make(rbd, model, B, dtype, layout, 141)
    shape = (B, 6 * model.nv) if layout == "aos" else (6 * model.nv, B)
    A = torch.zeros(shape, dtype=TD[dtype], device="cuda")
    rbd.momentum_matrix_(A, state)
    if "kin_spec" not in rbd.last_kernel(state):
        pytest.skip("no compiled kernel on this box: " + rbd.last_kernel(state))
    com = rbd.center_of_mass(state)
    ke, pe = rbd.kinetic_energy(state), rbd.gravitational_potential_energy(state)
    h, hb = rbd.momentum(state), rbd.momentum_rate_bias(state)
    assert "mom_spec" in rbd.last_kernel(state) or dtype == "f32"  # (a program whose registers spilled steps aside: the fp32 momentum walk of the larger trees)
    assert rbd.sync(state) == 0
    rel = lambda got, ref: np.abs(got - ref).max() / max(1.0, np.abs(ref).max())
    A_ref, _, com_ref = oracle.momentum_matrix(model, q, v)
    ke_ref, pe_ref = oracle.energy(model, q, v)
    h_ref, hb_ref = oracle.momentum(model, q, v)
    assert rel(host(A, state).reshape(B, model.nv, 6).transpose(0, 2, 1), A_ref) <= tol
    assert rel(host(com, state), com_ref) <= tol
    assert rel(ke.double().cpu().numpy(), ke_ref) <= tol and rel(pe.double().cpu().numpy(), pe_ref) <= tol
    assert rel(h.double().cpu().numpy(), h_ref) <= 10 * tol and rel(hb.double().cpu().numpy(), hb_ref) <= 100 * tol
    rng = np.random.default_rng(142)
    J = torch.full(shape, 7.0, dtype=TD[dtype], device="cuda")  # off-path columns must be zeroed
    for _ in range(5):
        base, body = rng.choice(np.arange(-1, model.n_bodies), 2, replace=False)
        rbd.geometric_jacobian_(J, state, int(base), int(body))
        assert "jac_spec" in rbd.last_kernel(state)
        ref, trel = oracle.geometric_jacobian(model, q, base, body, v)
        got = host(J, state).reshape(B, model.nv, 6).transpose(0, 2, 1)
        assert rel(got, ref) <= tol and rel(np.einsum("bkn,bn->bk", got, v), trel) <= 10 * tol


def _with_gains(model, gains):
    """a copy of the flat model whose loop joints carry `gains` (one 4-tuple per loop joint) — what the oracle reads"""
    import copy
    m = copy.copy(model)
    m.loops = [dict(l, gains=tuple(g)) for l, g in zip(model.loops, gains)]
    m._c = None
    return m


@pytest.mark.parametrize("kernels", ["compiled", "fused", "three_launches"])
def test_four_bar_custom_stabilization_gains(rbd, oracle, models, kernels, monkeypatch):
    """`stabilization_gains` as a per-call argument (src/mechanism_algorithms.jl:614-632, :848): non-default Baumgarte gains — one SE3PDGains for every
    loop joint (the ConstDict case) and a dict keyed by loop joint — through each of the three kernel forms of the small-loop branch, against the oracle
    evaluating a model that carries those gains; then back to the defaults on the same workspace.  Anything that is not a gains object raises."""
    monkeypatch.setenv("RBD_JIT", "1" if kernels == "compiled" else "0")
    if kernels == "three_launches":
        tune(monkeypatch, loop_no_fused="1")
    model = models["four_bar"]
    B = 512
    q, v, tau = four_bar_inputs(rbd, B, 77)
    state = rbd.MechanismState(model, B)
    result = rbd.DynamicsResult(model, B)
    rbd.set_configuration_(state, q)
    rbd.set_velocity_(state, v)
    custom = (400.0, 40.0, 250.0, 10.0)
    g = rbd.SE3PDGains(rbd.PDGains(custom[0], custom[1]), rbd.PDGains(custom[2], custom[3]))
    name = model.loops[0]["name"]
    refs = {}
    for label, arg, gains in (("default", "default", model.loops[0]["gains"]), ("const", g, custom), ("dict", {name: g}, custom), ("index", {0: g}, custom),
                              ("default again", "default", model.loops[0]["gains"]), ("default object", rbd.default_constraint_stabilization_gains(), (100.0, 20.0, 100.0, 20.0))):
        result.vd.fill_(float("nan"))
        rbd.dynamics_(result, state, dev(tau, state), stabilization_gains=arg)
        assert rbd.sync(state) == 0
        ref = oracle.dynamics_loops(_with_gains(model, [gains]), q, v, tau, stabilize=True)
        refs[label] = ref["vdot"]
        assert np.abs(host(result.vd, state) - ref["vdot"]).max() <= 1e-10 * max(1.0, np.abs(ref["vdot"]).max()), label
        assert np.abs(host(result.constraintbias, state) - ref["k"]).max() <= 1e-10 * max(1.0, np.abs(ref["k"]).max()), label
    assert np.abs(refs["const"] - refs["default"]).max() > 1e-3  # the gains matter on these states (joint 1 is off the closure)
    for bad in (3.0, (100.0, 20.0, 100.0, 20.0), "none", {name: (1.0, 2.0, 3.0, 4.0)}):
        with pytest.raises(ValueError):
            rbd.dynamics_(result, state, dev(tau, state), stabilization_gains=bad)
    with pytest.raises(KeyError):
        rbd.dynamics_(result, state, dev(tau, state), stabilization_gains={"no such joint": g})
    # simulate takes the same keyword (src/simulate.jl:37): two steps with the custom gains against the oracle's integrator on a model carrying them
    import simulate_np
    rbd.set_configuration_(state, q)
    rbd.set_velocity_(state, v)
    rbd.simulate_(state, 1.5e-3, dt=1e-3, stabilization_gains=g)
    _, q_ref, v_ref = simulate_np.simulate(_with_gains(model, [custom]), q[:4], v[:4], 1.5e-3, 1e-3, stabilize=True)
    assert np.abs(host(state.q, state)[:4] - q_ref).max() <= 1e-10 and np.abs(host(state.v, state)[:4] - v_ref).max() <= 1e-9


def test_maximal_coordinates_per_joint_gains(rbd, oracle):
    """Per-joint gains on a mechanism with several loop joints (the generic loop kernels): every loop joint its own SE3PDGains, states pushed off the
    constraint manifold so that the stabilization term is not zero."""
    from test_oracle_loops import MC_JOINTS, maximal_state
    rng = np.random.default_rng(54)
    tree = rbd.rand_tree_mechanism(rng, MC_JOINTS)
    mt, mc = rbd.flatten(tree), rbd.flatten(rbd.maximal_coordinates(tree))
    B = 9
    q, v = rbd.rand_configuration(mt, B, rng), rbd.rand_velocity(mt, B, rng)
    H, T, _ = oracle.body_kinematics(mt, q, v, np.zeros((B, mt.nv)))
    qm, vm = maximal_state(H, T)
    qm = (qm + 1e-3 * rng.standard_normal(qm.shape)).reshape(B, -1, 7)   # off the constraint manifold, in rotation and translation ...
    qm[:, :, :4] /= np.linalg.norm(qm[:, :, :4], axis=2, keepdims=True)       # ... with unit quaternions (non-unit ones are outside what the reference pins)
    qm = qm.reshape(B, -1)
    vm = vm + 1e-3 * rng.standard_normal(vm.shape)
    gains = [tuple(rng.uniform(10.0, 300.0, 4)) for _ in mc.loops]
    arg = {l["name"]: rbd.SE3PDGains(rbd.PDGains(g[0], g[1]), rbd.PDGains(g[2], g[3])) for l, g in zip(mc.loops, gains)}
    state = rbd.MechanismState(mc, B)
    result = rbd.DynamicsResult(mc, B)
    rbd.set_configuration_(state, qm)
    rbd.set_velocity_(state, vm)
    out = {}
    for label, a, gg in (("custom", arg, gains), ("default", "default", [l["gains"] for l in mc.loops])):
        rbd.dynamics_(result, state, stabilization_gains=a)
        assert rbd.sync(state) == 0
        ref = oracle.dynamics_loops(_with_gains(mc, gg), qm, vm)
        out[label] = ref["k"]
        assert np.abs(host(result.constraintbias, state) - ref["k"]).max() <= 1e-9 * max(1.0, np.abs(ref["k"]).max()), label
        assert np.abs(host(result.vd, state) - ref["vdot"]).max() <= 1e-8 * max(1.0, np.abs(ref["vdot"]).max()), label
    assert np.abs(out["custom"] - out["default"]).max() > 1e-3


@pytest.mark.gpu
def test_maximal_coordinates_loop_dynamics(rbd, oracle):
    """A mechanism in maximal coordinates (test/test_mechanism_modification.jl:274-318): 8 floating bodies, 8 loop joints of every
    supported type, nv = 48, nc = 34 — the loop branch of dynamics! at sizes where its work arrays fall back to the HBM scratch.
    v̇ and the constraint Jacobian / bias against the oracle; the Schur matrix is rank-deficient here, so λ is the minimum-norm one."""
    from test_oracle_loops import MC_JOINTS, maximal_state
    rng = np.random.default_rng(53)
    tree = rbd.rand_tree_mechanism(rng, MC_JOINTS)
    mt, mc = rbd.flatten(tree), rbd.flatten(rbd.maximal_coordinates(tree))
    B = 33
    q, v = rbd.rand_configuration(mt, B, rng), rbd.rand_velocity(mt, B, rng)
    H, T, _ = oracle.body_kinematics(mt, q, v, np.zeros((B, mt.nv)))
    qm, vm = maximal_state(H, T)
    tau = rng.random((B, mc.nv))
    state = rbd.MechanismState(mc, B)
    result = rbd.DynamicsResult(mc, B)
    rbd.set_configuration_(state, qm)
    rbd.set_velocity_(state, vm)
    rbd.dynamics_(result, state, dev(tau, state))
    assert rbd.sync(state) == 0
    ref = oracle.dynamics_loops(mc, qm, vm, tau)
    got = host(result.vd, state)
    assert np.abs(got - ref["vdot"]).max() <= 1e-8 * max(1.0, np.abs(ref["vdot"]).max())
    K = host(result.constraintjacobian, state).reshape(B, mc.nv, mc.nc).transpose(0, 2, 1)
    assert np.abs(K - ref["K"]).max() <= 1e-11
    assert np.abs(host(result.constraintbias, state) - ref["k"]).max() <= 1e-9 * max(1.0, np.abs(ref["k"]).max())
    # KKT residuals at the GPU solution (independent of the λ choice)
    Mg = host(result.massmatrix, state).reshape(B, mc.nv, mc.nv).transpose(0, 2, 1)
    Ms = np.tril(Mg) + np.transpose(np.tril(Mg, -1), (0, 2, 1))
    lam = host(result.lambda_, state)
    r1 = np.einsum("bij,bj->bi", Ms, got) + host(result.dynamicsbias, state) + np.einsum("bcv,bc->bv", K, lam) - tau
    assert np.abs(r1).max() <= 1e-8 * max(1.0, np.abs(tau).max())


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_trees_of_more_than_64_bodies(rbd, oracle, dtype):
    """The reference has no size limit (its `rand_chain_mechanism`, src/mechanism_modification.jl:402, is used with 100 joints); the wavefront-shaped
    kernels stop at 64 bodies.  Beyond that the any-size kernels of rbd_big_kernels.hip (one thread per state, HBM scratch) run the reference's own route:
    a 100-joint chain and a 150-body random tree of every joint type — inverse_dynamics! with its per-body outputs, dynamics_bias!, mass_matrix!,
    mass_matrix_solve and dynamics! against the oracle; entry points outside the four hot-path functions refuse such a model."""
    rng = np.random.default_rng(402)
    chain = ["QuaternionFloating"] + ["Revolute"] * 60 + ["Prismatic"] * 15 + ["SinCosRevolute"] * 10 + ["Fixed"] * 6 + ["Planar"] * 4 + ["QuaternionSpherical"] * 4
    tree = ["QuaternionFloating"] + ["Revolute"] * 100 + ["Prismatic"] * 20 + ["Fixed"] * 10 + ["QuaternionSpherical"] * 9 + ["Planar"] * 10
    for joints, sel in ((chain, lambda m, r: m.bodies[-1]), (tree, None)):
        order = rng.permutation(len(joints) - 1)
        joints = [joints[0]] + [joints[1 + k] for k in order]
        model = rbd.flatten(rbd.rand_tree_mechanism(rng, joints, sel))
        assert model.n_bodies == len(joints) > 64
        B = 70 if dtype == "f64" else 33
        state, q, v, vd, fe = make(rbd, model, B, dtype, "aos", 403)
        nb = model.n_bodies
        rt = 1e-10 if dtype == "f64" else 5e-4
        rel = lambda a, b: np.abs(a - b).max() / max(1.0, np.abs(b).max())
        tau = torch.zeros_like(state.v)
        jw = torch.full_like(dev(fe, state), float("nan"))
        acc = torch.full_like(jw, float("nan"))
        rbd.inverse_dynamics_(tau, state, dev(vd, state), dev(fe, state), jointwrenchesout=jw, accelerations=acc)
        t_ref, jw_ref, acc_ref = oracle.inverse_dynamics_bodies(model, q, v, vd, fe)
        assert rel(host(tau, state), t_ref) <= rt and rel(host(jw, state).reshape(B, nb, 6), jw_ref) <= rt and rel(host(acc, state).reshape(B, nb, 6), acc_ref) <= rt
        result = rbd.DynamicsResult(model, B, dtype=TD[dtype])
        rbd.dynamics_bias_(result, state, dev(fe, state))
        assert rel(host(result.dynamicsbias, state), oracle.dynamics_bias(model, q, v, fe)) <= rt
        rbd.mass_matrix_(result, state)
        Mg = host(result.massmatrix, state).reshape(B, model.nv, model.nv).transpose(0, 2, 1)
        Mref = oracle.mass_matrix(model, q)
        assert rel(np.tril(Mg), np.tril(Mref)) <= rt
        if dtype == "f64":  # the dense solve of a 100-dof chain: hold it to the backward error of M v̇ = τ − c, and to the oracle up to the conditioning
            rbd.dynamics_(result, state, dev(vd, state), dev(fe, state))
            assert rbd.sync(state) == 0
            got = host(result.vd, state)
            Ms = np.tril(Mref) + np.transpose(np.tril(Mref, -1), (0, 2, 1))
            rhs = vd - oracle.dynamics_bias(model, q, v, fe)
            res = np.einsum("bij,bj->bi", Ms, got) - rhs
            eta = np.linalg.norm(res, axis=1) / (np.linalg.norm(Ms, axis=(1, 2)) * np.linalg.norm(got, axis=1) + np.linalg.norm(rhs, axis=1))
            assert eta.max() <= 1e-13, eta.max()
            ref = oracle.dynamics(model, q, v, vd, fe)
            cond = np.linalg.cond(Ms).max()
            assert rel(got, ref) <= 1e-14 * cond, (rel(got, ref), cond)
            x = torch.zeros_like(state.v)
            Mo = torch.full_like(result.massmatrix, float("nan"))
            rbd.mass_matrix_solve_(x, state, dev(vd, state), M_out=Mo)
            xr = np.linalg.solve(Ms, vd[..., None])[..., 0]
            assert rel(host(x, state), xr) <= 1e-14 * cond
            # M_out and result.massmatrix hold M after the solve, not its Cholesky factor: the reference factors a copy (result.L, :763-764)
            low = lambda t: np.tril(host(t, state).reshape(B, model.nv, model.nv).transpose(0, 2, 1))
            assert rel(low(Mo), np.tril(Mref)) <= rt
            result.massmatrix.fill_(float("nan"))
            rbd.dynamics_(result, state, dev(vd, state), dev(fe, state), algorithm="crba")
            assert rbd.sync(state) == 0
            assert rel(low(result.massmatrix), np.tril(Mref)) <= rt and rel(host(result.vd, state), ref) <= 1e-14 * cond
            assert rel(host(result.dynamicsbias, state), oracle.dynamics_bias(model, q, v, fe)) <= rt


@pytest.mark.gpu
def test_simulate_on_a_tree_of_more_than_64_bodies(rbd, oracle):
    """Round 5: `simulate` has no size limit in the reference (src/simulate.jl:36-55).  A 70-body random tree of every joint type: three RK4 steps — the stage in
    launches of its own over the any-size tables (big_mk_stage_kernel) around rbd_dynamics' any-size route — against the numpy restatement of the integrator;
    and the stage-by-stage form a host-side controller uses (rbd_mk_stage)."""
    import simulate_np
    rng = np.random.default_rng(64)
    joints = ["QuaternionFloating"] + ["Revolute"] * 50 + ["Prismatic"] * 8 + ["SinCosRevolute"] * 4 + ["Fixed"] * 3 + ["Planar"] * 2 + ["QuaternionSpherical"] * 2
    order = rng.permutation(len(joints) - 1)
    model = rbd.flatten(rbd.rand_tree_mechanism(rng, [joints[0]] + [joints[1 + k] for k in order]))
    assert model.n_bodies == 70
    B, dt = 5, 2e-3
    q0, v0 = rbd.rand_configuration(model, B, rng), 0.2 * rbd.rand_velocity(model, B, rng)
    tau = 0.1 * rng.standard_normal((B, model.nv))
    state = rbd.MechanismState(model, B)
    rbd.set_configuration_(state, q0)
    rbd.set_velocity_(state, v0)
    ts = rbd.simulate_(state, 3 * dt - 1e-12, dt=dt, torques=dev(tau, state))
    assert len(ts) == 4
    _, q_ref, v_ref = simulate_np.simulate(model, q0, v0, 3 * dt - 1e-12, dt, tau=tau)
    qg, vg = host(state.q, state), host(state.v, state)
    assert np.abs(canon_q(model, qg) - canon_q(model, q_ref)).max() <= 1e-10 * max(1.0, np.abs(q_ref).max())
    assert np.abs(vg - v_ref).max() <= 1e-8 * max(1.0, np.abs(v_ref).max())
    # the same three steps with a controller on the host: control_(torques, t, state) before every stage
    rbd.set_configuration_(state, q0)
    rbd.set_velocity_(state, v0)
    calls = []
    def control_(torques, t, st):
        calls.append(t)
        torques.copy_(dev(tau, st))
    rbd.simulate_(state, 3 * dt - 1e-12, control_, dt=dt)
    assert len(calls) == 12
    assert np.abs(canon_q(model, host(state.q, state)) - canon_q(model, q_ref)).max() <= 1e-10 * max(1.0, np.abs(q_ref).max())
    assert np.abs(host(state.v, state) - v_ref).max() <= 1e-8 * max(1.0, np.abs(v_ref).max())


def _big_tree(rbd, seed=64):
    rng = np.random.default_rng(seed)
    joints = ["QuaternionFloating"] + ["Revolute"] * 50 + ["Prismatic"] * 8 + ["SinCosRevolute"] * 4 + ["Fixed"] * 3 + ["Planar"] * 2 + ["QuaternionSpherical"] * 2
    order = rng.permutation(len(joints) - 1)
    model = rbd.flatten(rbd.rand_tree_mechanism(rng, [joints[0]] + [joints[1 + k] for k in order]))
    assert model.n_bodies == 70
    return model, rng


def _star(rbd, seed=65, spokes=11):
    """a hub with more children than a lane-per-body record lists (IB_MAXCHILD = 8): eleven 1-dof / 3-dof spokes on a floating hub, two of them two bodies long"""
    rng = np.random.default_rng(seed)
    kinds = ["Revolute", "Prismatic", "Planar", "QuaternionSpherical", "SinCosRevolute", "Revolute", "Revolute", "Prismatic", "Revolute", "Fixed", "Revolute"][:spokes]
    spec = [("QuaternionFloating", [(k, [("Revolute", [])] if i < 2 else []) for i, k in enumerate(kinds)])]
    return rbd.flatten(rbd.builders.tree_mechanism(rng, spec)), rng


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["aos", "soa"])
@pytest.mark.parametrize("tree", ["70_bodies", "11_children"])
def test_kinematics_byproducts_without_a_size_limit(rbd, oracle, tree, layout):
    """Round 6: momentum_matrix!, center_of_mass, the energies, geometric_jacobian!, momentum and momentum_rate_bias have no size limit in the reference
    (src/mechanism_algorithms.jl:28-50, :80-99, :313-327; src/mechanism_state.jl:886-903, :975-987).  A 70-body random tree of every joint type, and a hub with
    eleven children (both outside the lane-per-body tables: one thread per state over the any-size tables, big_kin_kernel) against the oracle, fp64 at 1e-11 and
    fp32."""
    model, rng = _big_tree(rbd) if tree == "70_bodies" else _star(rbd)
    assert tree == "70_bodies" or max(np.bincount(np.asarray(model.parent)[np.asarray(model.parent) >= 0])) > 8
    B = 37
    for dtype, tol in (("f64", 1e-11), ("f32", 3e-4)):
        state, q, v, _, _ = make(rbd, model, B, dtype, layout, 131)
        A = torch.zeros_like(state.v).new_zeros((B, 6 * model.nv) if layout == "aos" else (6 * model.nv, B))
        rbd.momentum_matrix_(A, state)
        com = rbd.center_of_mass(state)
        ke, pe = rbd.kinetic_energy(state), rbd.gravitational_potential_energy(state)
        h, hb = rbd.momentum(state), rbd.momentum_rate_bias(state)
        assert rbd.sync(state) == 0
        rel = lambda got, ref: np.abs(got - ref).max() / max(1.0, np.abs(ref).max())
        A_ref, _, com_ref = oracle.momentum_matrix(model, q, v)
        ke_ref, pe_ref = oracle.energy(model, q, v)
        h_ref, hb_ref = oracle.momentum(model, q, v)
        assert rel(host(A, state).reshape(B, model.nv, 6).transpose(0, 2, 1), A_ref) <= tol
        assert rel(host(com, state), com_ref) <= tol
        assert rel(ke.double().cpu().numpy(), ke_ref) <= tol and rel(pe.double().cpu().numpy(), pe_ref) <= tol
        assert rel(h.double().cpu().numpy(), h_ref) <= tol and rel(hb.double().cpu().numpy(), hb_ref) <= 10 * tol
        J = torch.full_like(A, 7.0)  # off-path columns must be zeroed
        for _ in range(5):
            base, body = rng.choice(np.arange(-1, model.n_bodies), 2, replace=False)
            rbd.geometric_jacobian_(J, state, int(base), int(body))
            ref, trel = oracle.geometric_jacobian(model, q, base, body, v)
            got = host(J, state).reshape(B, model.nv, 6).transpose(0, 2, 1)
            assert rel(got, ref) <= tol and rel(np.einsum("bkn,bn->bk", got, v), trel) <= 10 * tol


@pytest.mark.gpu
@pytest.mark.parametrize("tree", ["70_bodies", "11_children"])
def test_dynamics_and_pd_controller_without_a_size_limit(rbd, oracle, tree):
    """... and the hot path itself on the hub with eleven children (refused until round 6), plus the device-side PD law of `simulate(state, T, control!)` on both
    trees (RBD_CONTROL_PD: refused for trees of more than 64 bodies until round 6): two RK4 steps against the numpy integrator with the same law as control!."""
    import simulate_np
    model, rng = _big_tree(rbd, 66) if tree == "70_bodies" else _star(rbd, 67)
    B, dt = 4, 1e-3
    state, q, v, tau, fe = make(rbd, model, B, "f64", "aos", 132, fext=True)
    result = rbd.DynamicsResult(model, B)
    rbd.dynamics_(result, state, dev(tau, state), dev(fe, state))
    assert rbd.sync(state) == 0
    ref = oracle.dynamics(model, q, v, tau, fe)
    assert np.abs(host(result.vd, state) - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max())
    out = torch.zeros_like(state.v)
    rbd.inverse_dynamics_(out, state, dev(ref, state), dev(fe, state))
    assert np.abs(host(out, state) - tau).max() <= 1e-9 * max(1.0, np.abs(tau).max())
    kp, kd = 2 * rng.random(model.nv), 0.02 * rng.random(model.nv)
    qdes, tff = 0.3 * rng.standard_normal((B, model.nq)), 0.1 * rng.random((B, model.nv))
    v0 = 0.2 * v
    rbd.set_velocity_(state, v0)
    rbd.simulate_(state, 1.5 * dt, control_=rbd.PDControl(torch.as_tensor(kp), torch.as_tensor(kd), dev(qdes, state), dev(tff, state)), dt=dt)
    nvj = {0: 0, 3: 6, 4: 3, 5: 3}
    one = np.array([1.0 if int(t) in (1, 2) else 0.0 for t in model.joint_type for _ in range(nvj.get(int(t), 1))])
    qidx = np.array([int(model.q_offset[b]) for b in range(model.n_bodies) for _ in range(nvj.get(int(model.joint_type[b]), 1))])
    pd = lambda b, t, qq, vv: tff[b] - one * (kp * (qq[qidx] - qdes[b][qidx]) + kd * vv)
    _, q_ref, v_ref = simulate_np.simulate(model, q, v0, 1.5 * dt, dt, control=pd)
    assert np.abs(canon_q(model, host(state.q, state)) - canon_q(model, q_ref)).max() <= 1e-10 * max(1.0, np.abs(q_ref).max())
    assert np.abs(host(state.v, state) - v_ref).max() <= 1e-8 * max(1.0, np.abs(v_ref).max())


@pytest.mark.gpu
def test_loop_joints_on_a_tree_of_more_than_64_bodies(rbd, oracle):
    """Round 4: the reference has no size limit for mechanisms with loop joints either (`constraint_jacobian!`, `constraint_bias!`, the loop branch of
    `dynamics_solve!`: src/mechanism_algorithms.jl:574-673, :768-816).  A 72-body random tree (floating base, revolute / prismatic / fixed joints) closed by three
    loop joints — Revolute, Fixed, QuaternionSpherical: nc = 14 — through the any-size kernels (bias forces, per-body kinematics, mass matrix) and the same
    constrained solve: v̇, K, k, M, c against the oracle, the KKT residuals, and per-call stabilization gains."""
    from rigidbodydynamics_jl_amd.builders import rand_joint_type, _rand_rotation
    from rigidbodydynamics_jl_amd.mechanism import Joint, Transform3D, attach_
    rng = np.random.default_rng(5)
    tree = rbd.rand_tree_mechanism(rng, ["QuaternionFloating"] + ["Revolute"] * 60 + ["Prismatic"] * 8 + ["Fixed"] * 3)
    bodies = tree.bodies[1:]
    for k, name in enumerate(["Revolute", "Fixed", "QuaternionSpherical"]):
        a, b = rng.choice(len(bodies), 2, replace=False)
        j = Joint(f"loop{k}", rand_joint_type(rng, name))
        attach_(tree, bodies[a], bodies[b], j, joint_pose=Transform3D(j.frame_before, bodies[a].default_frame, _rand_rotation(rng), rng.uniform(-0.5, 0.5, 3)),
                successor_pose=Transform3D(bodies[b].default_frame, j.frame_after, _rand_rotation(rng), rng.uniform(-0.5, 0.5, 3)))
    model = rbd.flatten(tree)
    assert model.n_bodies == 72 and model.n_loops == 3 and model.nc == 14
    B = 37
    q, v, tau = rbd.rand_configuration(model, B, rng), rbd.rand_velocity(model, B, rng), rng.random((B, model.nv))
    state = rbd.MechanismState(model, B)
    result = rbd.DynamicsResult(model, B)
    rbd.set_configuration_(state, q)
    rbd.set_velocity_(state, v)
    custom = rbd.SE3PDGains(rbd.PDGains(30.0, 7.0), rbd.PDGains(55.0, 9.0))
    for gains_arg, gains in (("default", [l["gains"] for l in model.loops]), (custom, [custom.as_tuple()] * 3), (None, None)):
        rbd.dynamics_(result, state, dev(tau, state), stabilization_gains=gains_arg)
        assert rbd.sync(state) == 0 and "big_" in rbd.last_kernel(state)
        ref = oracle.dynamics_loops(_with_gains(model, gains) if gains else model, q, v, tau, stabilize=gains is not None)
        got = host(result.vd, state)
        assert np.abs(got - ref["vdot"]).max() <= 1e-8 * max(1.0, np.abs(ref["vdot"]).max()), gains_arg
        K = host(result.constraintjacobian, state).reshape(B, model.nv, model.nc).transpose(0, 2, 1)
        assert np.abs(K - ref["K"]).max() <= 1e-11 * max(1.0, np.abs(ref["K"]).max())
        assert np.abs(host(result.constraintbias, state) - ref["k"]).max() <= 1e-9 * max(1.0, np.abs(ref["k"]).max())
    Mg = host(result.massmatrix, state).reshape(B, model.nv, model.nv).transpose(0, 2, 1)
    Mref = oracle.mass_matrix(model, q)
    assert np.abs(np.tril(Mg) - np.tril(Mref)).max() <= 1e-10 * max(1.0, np.abs(Mref).max())
    assert np.abs(host(result.dynamicsbias, state) - oracle.dynamics_bias(model, q, v, None)).max() <= 1e-9 * max(1.0, np.abs(host(result.dynamicsbias, state)).max())
    Ms = np.tril(Mg) + np.transpose(np.tril(Mg, -1), (0, 2, 1))
    lam = host(result.lambda_, state)
    r1 = np.einsum("bij,bj->bi", Ms, got) + host(result.dynamicsbias, state) + np.einsum("bcv,bc->bv", K, lam) - tau
    assert np.abs(r1).max() <= 1e-8 * max(1.0, np.abs(host(result.dynamicsbias, state)).max())
    # ... and `simulate` (round 5: the integrator's stage over the any-size tables): one RK4 step, loop joints and their stabilization included
    import simulate_np
    rbd.simulate_(state, 1e-3 - 1e-12, dt=1e-3, torques=dev(tau, state))
    n = 3
    _, q_ref, v_ref = simulate_np.simulate(model, q[:n], v[:n], 1e-3 - 1e-12, 1e-3, tau=tau[:n])
    assert np.abs(canon_q(model, host(state.q, state)[:n]) - canon_q(model, q_ref)).max() <= 1e-9 * max(1.0, np.abs(q_ref).max())
    assert np.abs(host(state.v, state)[:n] - v_ref).max() <= 1e-7 * max(1.0, np.abs(v_ref).max())
    rbd.set_configuration_(state, q)
    rbd.set_velocity_(state, v)
    tv = torch.zeros_like(state.v)
    with pytest.raises(Exception):  # inverse_dynamics! on a mechanism with loop joints: "can currently only handle tree Mechanisms" (:549)
        rbd.inverse_dynamics_(tv, state, dev(tau, state))


@pytest.mark.gpu
def test_maximal_coordinates_at_the_reference_size(rbd, oracle):
    """The reference's own pin of constraint_jacobian! / constraint_bias! / the loop branch of dynamics_solve! at ITS size
    (test/test_mechanism_modification.jl:274-318): QuaternionFloating + 10 Revolute + QuaternionSpherical + Planar + 10 Fixed + 5 SinCosRevolute
    = 28 joints, every body on a floating joint in maximal coordinates — nv = 168 (three mask words per mass-matrix row), nc = 141.  The body
    accelerations of the tree and of the maximal-coordinates mechanism agree (the reference's assertion, atol 1e-10 there), and v̇, K, k match the oracle."""
    rng = np.random.default_rng(53)
    joints = ["QuaternionFloating"] + ["Revolute"] * 10 + ["QuaternionSpherical", "Planar"] + ["Fixed"] * 10 + ["SinCosRevolute"] * 5
    tree = rbd.rand_tree_mechanism(rng, joints)
    mt, mc = rbd.flatten(tree), rbd.flatten(rbd.maximal_coordinates(tree))
    assert mc.n_bodies == 28 and mc.nv == 168 and mc.n_loops == 28
    from test_oracle_loops import maximal_state
    B = 6
    q, v = rbd.rand_configuration(mt, B, rng), rbd.rand_velocity(mt, B, rng)
    vd_tree = oracle.dynamics(mt, q, v)
    H, T, A_tree = oracle.body_kinematics(mt, q, v, vd_tree)
    qm, vm = maximal_state(H, T)
    state = rbd.MechanismState(mc, B)
    result = rbd.DynamicsResult(mc, B)
    rbd.set_configuration_(state, qm)
    rbd.set_velocity_(state, vm)
    rbd.dynamics_(result, state)
    assert rbd.sync(state) == 0
    got = host(result.vd, state)
    ref = oracle.dynamics_loops(mc, qm, vm)
    assert np.abs(got - ref["vdot"]).max() <= 1e-8 * max(1.0, np.abs(ref["vdot"]).max())
    K = host(result.constraintjacobian, state).reshape(B, mc.nv, mc.nc).transpose(0, 2, 1)
    assert np.abs(K - ref["K"]).max() <= 1e-11
    assert np.abs(host(result.constraintbias, state) - ref["k"]).max() <= 1e-9 * max(1.0, np.abs(ref["k"]).max())
    _, _, A_mc = oracle.body_kinematics(mc, qm, vm, got)  # body accelerations with the GPU's v̇
    assert np.abs(A_mc - A_tree).max() <= 1e-8 * max(1.0, np.abs(A_tree).max())
    # the mass matrix with its structural zeros (three 64-bit mask words per row)
    Mg = host(result.massmatrix, state).reshape(B, mc.nv, mc.nv).transpose(0, 2, 1)
    Mref = oracle.mass_matrix(mc, qm)
    assert np.abs(np.tril(Mg) - np.tril(Mref)).max() <= 1e-10 * max(1.0, np.abs(Mref).max())


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["aos", "soa"])
@pytest.mark.parametrize("name", MODELS)
def test_momentum_and_rate_bias_f64(rbd, oracle, models, name, layout):
    """momentum(state), momentum_rate_bias(state) (src/mechanism_state.jl:975-987) against the oracle; and momentum = A v."""
    model = models[name]
    B = 41
    state, q, v, _, _ = make(rbd, model, B, "f64", layout, 95)
    h, hb = rbd.momentum(state), rbd.momentum_rate_bias(state)
    torch.cuda.synchronize()
    h_ref, hb_ref = oracle.momentum(model, q, v)
    assert np.abs(h.cpu().numpy() - h_ref).max() <= 1e-11 * max(1.0, np.abs(h_ref).max())
    assert np.abs(hb.cpu().numpy() - hb_ref).max() <= 1e-10 * max(1.0, np.abs(hb_ref).max())
    A = torch.zeros((B, 6 * model.nv) if layout == "aos" else (6 * model.nv, B), dtype=torch.float64, device="cuda")
    rbd.momentum_matrix_(A, state)
    Av = np.einsum("bkn,bn->bk", host(A, state).reshape(B, model.nv, 6).transpose(0, 2, 1), v)
    assert np.abs(Av - h.cpu().numpy()).max() <= 1e-11 * max(1.0, np.abs(Av).max())


@pytest.mark.gpu
@pytest.mark.parametrize("jit", JIT)
@pytest.mark.parametrize("name", ["atlas_floating", "atlas_fixed", "valkyrie_floating"])
def test_simulate_banked_fused_path_matches_oracle(rbd, oracle, models, name, jit, monkeypatch):
    monkeypatch.setenv("RBD_JIT", "1" if jit == "compiled" else "0")
    """`simulate` through the two-bodies-per-lane kernel with the integrator stage fused in (what large batches run): forced at a
    small batch with RBD_TUNE bank_min_batch so that every state can be compared with the numpy restatement of the Munthe-Kaas step."""
    import simulate_np
    tune(monkeypatch, bank_min_batch="1", sim_walk_min_batch=1 << 40)  # read when the workspace is created (round 6: without the second knob `simulate` takes the looped walk program at every batch)
    model = models[name]
    B, dt, T = 5, 1e-3, 0.0045  # 5 steps: first (stage 0 alone), middle ones (previous step closed inside stage 0), closing launch
    q, v, tau, _ = rand_inputs(rbd, model, B, 97, fext=True)
    state = rbd.MechanismState(model, B)
    rbd.set_configuration_(state, q)
    rbd.set_velocity_(state, v)
    rbd.simulate_(state, T, dt=dt, torques=dev(tau, state))
    from rigidbodydynamics_jl_amd import _capi
    k = _capi.lib().rbd_workspace_last_kernel(state.ws.handle)
    assert k.startswith(b"aba_bank_kernel") and (b"compiled" not in k or jit == "compiled"), k
    _, q_ref, v_ref = simulate_np.simulate(model, q, v, T, dt, tau)
    qg, vg = host(state.q, state), host(state.v, state)
    assert np.abs(canon_q(model, qg) - canon_q(model, q_ref)).max() <= 1e-11 * max(1.0, np.abs(q_ref).max())
    assert np.abs(vg - v_ref).max() <= 1e-9 * max(1.0, np.abs(v_ref).max())


@pytest.mark.gpu
def test_mechanism_without_degrees_of_freedom(rbd, oracle):
    """All tree joints Fixed: nq = nv = 0 — every entry point is a no-op on empty buffers, as with the reference's empty vectors."""
    rng = np.random.default_rng(3)
    model = rbd.flatten(rbd.rand_tree_mechanism(rng, ["Fixed"] * 3))
    assert (model.nq, model.nv) == (0, 0)
    B = 5
    state = rbd.MechanismState(model, B)
    result = rbd.DynamicsResult(model, B)
    rbd.dynamics_(result, state)
    out = torch.zeros((B, 0), dtype=torch.float64, device="cuda")
    rbd.inverse_dynamics_(out, state, out)
    rbd.dynamics_bias_(out, state)
    torch.cuda.synchronize()
    assert result.vd.shape == (B, 0)
    com = rbd.center_of_mass(state)
    _, _, com_ref = oracle.momentum_matrix(model, np.zeros((B, 0)), np.zeros((B, 0)))
    assert np.abs(com.cpu().numpy() - com_ref).max() <= 1e-13


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("name", ["atlas_floating", "atlas_fixed", "double_pendulum", "randmech1"])
def test_batch_states_are_isolated_from_a_nan_state(rbd, models, name, dtype):
    """The reference evaluates one state per call; here several states share a wavefront.  A state with NaN / Inf inputs (a diverged
    trajectory) must not change its neighbours by one bit: every entry point and every lane mapping, against a clean run."""
    model = models[name]
    B = 37
    state, q, v, tau, fe = make(rbd, model, B, dtype, "aos", 123)
    t, f = dev(tau, state), dev(fe, state)
    algos = ["aba_lanes"] + [a for a in ("aba_banks", "aba_walk") if name != "randmech1" and (a != "aba_banks" or rbd.bank_plan(model))]

    def run_all():
        out = {}
        for a in algos:
            r = rbd.DynamicsResult(model, B, dtype=TD[dtype])
            rbd.dynamics_(r, state, t, f, algorithm=a)
            out[a] = r.vd.clone()
        tau_out = torch.zeros_like(state.v)
        rbd.inverse_dynamics_(tau_out, state, t, f)
        out["rnea"] = tau_out
        r = rbd.DynamicsResult(model, B, dtype=TD[dtype])
        rbd.mass_matrix_(r, state)
        out["crba"] = torch.tril(r.massmatrix.reshape(B, model.nv, model.nv).transpose(1, 2)).clone()
        torch.cuda.synchronize()
        return out

    clean = run_all()
    bad = [5, 20]  # neighbours inside a wavefront for every states-per-wave of these models
    state.q[bad[0]] = float("nan")
    state.v[bad[1]] = float("inf")
    dirty = run_all()
    ok = torch.ones(B, dtype=torch.bool)
    ok[bad] = False
    for k in clean:
        a, b = clean[k][ok.to(clean[k].device)], dirty[k][ok.to(clean[k].device)]
        assert torch.equal(a, b), (k, float((a - b).abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["aos", "soa"])
@pytest.mark.parametrize("name", MODELS)
def test_per_body_outputs_of_inverse_dynamics_and_dynamics_f64(rbd, oracle, models, name, layout):
    """The per-body results the reference leaves beside the torques: jointwrenchesout / accelerations of inverse_dynamics!
    (src/mechanism_algorithms.jl:542-553) and result.accelerations / jointwrenches / totalwrenches after dynamics! (:851-856,
    src/dynamics_result.jl:26-29), root frame, reference body order, at the reference's own atol 1e-10."""
    model = models[name]
    B, nb = 21, model.n_bodies
    state, q, v, vd, fe = make(rbd, model, B, "f64", layout, 131)
    tau = torch.zeros_like(state.v)
    jw = torch.full_like(dev(fe, state), float("nan"))
    acc = torch.full_like(jw, float("nan"))
    rbd.inverse_dynamics_(tau, state, dev(vd, state), dev(fe, state), jointwrenchesout=jw, accelerations=acc)
    t_ref, jw_ref, acc_ref = oracle.inverse_dynamics_bodies(model, q, v, vd, fe)
    tol = lambda r: 1e-10 * max(1.0, np.abs(r).max())
    assert np.abs(host(tau, state) - t_ref).max() <= tol(t_ref)
    assert np.abs(host(jw, state).reshape(B, nb, 6) - jw_ref).max() <= tol(jw_ref)
    assert np.abs(host(acc, state).reshape(B, nb, 6) - acc_ref).max() <= tol(acc_ref)
    # dynamics!: v̇ as before, and the per-body fields hold what its dynamics_bias! call computed, totalwrenches = the external wrenches
    for algorithm in ("aba", "crba"):
        result = rbd.DynamicsResult(model, B, layout=layout, bodies=True)
        rbd.dynamics_(result, state, dev(vd, state), dev(fe, state), algorithm=algorithm)
        c_ref, jw_ref, acc_ref = oracle.inverse_dynamics_bodies(model, q, v, None, fe)
        assert np.abs(host(result.vd, state) - oracle.dynamics(model, q, v, vd, fe)).max() <= 1e-10 * max(1.0, np.abs(oracle.dynamics(model, q, v, vd, fe)).max())
        assert np.abs(host(result.dynamicsbias, state) - c_ref).max() <= tol(c_ref)
        assert np.abs(host(result.jointwrenches, state).reshape(B, nb, 6) - jw_ref).max() <= tol(jw_ref)
        assert np.abs(host(result.accelerations, state).reshape(B, nb, 6) - acc_ref).max() <= tol(acc_ref)
        assert np.array_equal(host(result.totalwrenches, state), fe)
    result = rbd.DynamicsResult(model, B, layout=layout, bodies=True)
    rbd.dynamics_(result, state)  # no external wrenches: totalwrenches are zero
    assert float(result.totalwrenches.abs().max()) == 0.0 if nb else True


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("mapping", ["lanes", "banks", "walk", "compiled"])
def test_per_body_outputs_from_every_mapping(rbd, oracle, models, mapping, dtype):
    """inverse_dynamics!(τ, jointwrenches, accelerations, …) — the call the reference's own benchmark makes (perf/runbenchmarks.jl:49-57) — from the
    one-body-per-lane, two-bodies-per-lane and walk kernels (fp32: two states per lane) and from the kernel compiled for the mechanism, ragged batches,
    with and without v̇ / wrenches."""
    model = models["atlas_floating"]
    nb = model.n_bodies
    for B, layout in ((131, "aos"), (259, "soa")):
        state, q, v, vd, fe = make(rbd, model, B, dtype, layout, 140 + B)
        tau = torch.zeros_like(state.v)
        jw = torch.full_like(dev(fe, state), float("nan"))
        acc = torch.full_like(jw, float("nan"))
        rbd.inverse_dynamics_(tau, state, dev(vd, state), dev(fe, state), mapping=mapping, jointwrenchesout=jw, accelerations=acc)
        t_ref, jw_ref, acc_ref = oracle.inverse_dynamics_bodies(model, q, v, vd, fe)
        rt = 1e-10 if dtype == "f64" else 2e-4
        tol = lambda r: rt * max(1.0, np.abs(r).max())
        assert np.abs(host(tau, state) - t_ref).max() <= tol(t_ref)
        assert np.abs(host(jw, state).reshape(B, nb, 6) - jw_ref).max() <= tol(jw_ref)
        assert np.abs(host(acc, state).reshape(B, nb, 6) - acc_ref).max() <= tol(acc_ref)
        if mapping == "compiled":  # fp32, state-major buffers: the lanes store batch-innermost scratch, a second kernel moves it
            assert ("rows_to_state_major" in rbd.last_kernel(state)) == (dtype == "f32" and layout == "aos")
        # only one of the two, no v̇ (dynamics_bias!), no wrenches
        acc2 = torch.full_like(jw, float("nan"))
        rbd.inverse_dynamics_(tau, state, torch.zeros_like(state.v), None, mapping=mapping, accelerations=acc2)
        t_ref, jw_ref, acc_ref = oracle.inverse_dynamics_bodies(model, q, v, None, None)
        assert np.abs(host(tau, state) - t_ref).max() <= tol(t_ref)
        assert np.abs(host(acc2, state).reshape(B, nb, 6) - acc_ref).max() <= tol(acc_ref)


@pytest.mark.gpu
def test_result_must_match_state(rbd, models):
    """A DynamicsResult of another batch size / dtype / layout is refused (the reference would raise DimensionMismatch) instead of being
    written out of bounds."""
    model = models["atlas_floating"]
    state, *_ = make(rbd, model, 8, "f64", "aos", 140)
    for bad in (rbd.DynamicsResult(model), rbd.DynamicsResult(model, 8, dtype=torch.float32), rbd.DynamicsResult(model, 8, layout="soa")):
        with pytest.raises((rbd.DimensionMismatch, ValueError)):
            rbd.dynamics_(bad, state)
    good = rbd.DynamicsResult(model, 8)
    good.vd = torch.zeros(4, model.nv, dtype=torch.float64, device="cuda")
    with pytest.raises(rbd.DimensionMismatch):
        rbd.dynamics_(good, state)


@pytest.mark.gpu
def test_c_abi_rccl_gather_single_rank(rbd, models):
    """rbd_comm_* / rbd_gather (the exchange step of the sharded batch through the C ABI, RCCL opened by the library itself): a world of
    one rank — all that a 1-GPU box can hold — must hand back the shard, as an all-gather and as a gather to rank 0."""
    model = models["atlas_floating"]
    B = 257
    state, q, v, tau, _ = make(rbd, model, B, "f64", "aos", 150)
    result = rbd.DynamicsResult(model, B)
    rbd.dynamics_(result, state, dev(tau, state))
    comm = rbd.Comm(rbd.Comm.unique_id(), 1, 0, 0)
    try:
        for root in (None, 0):
            out = comm.gather(result.vd, root)
            torch.cuda.synchronize()
            assert torch.equal(out, result.vd)
        out32 = comm.gather(result.vd.float())
        torch.cuda.synchronize()
        assert torch.equal(out32, result.vd.float())
        # ... and the ragged form (rbd_gatherv: a count per rank — here one rank; more ranks need more GPUs than the test box has)
        for root in (None, 0):
            out = comm.gatherv(result.vd, [B], root)
            torch.cuda.synchronize()
            assert torch.equal(out, result.vd)
        with pytest.raises(ValueError):
            comm.gatherv(result.vd, [B - 1])
        # RBD_COMM_CHECK=1 (round 6): the ranks compare their counts before the grouped sends — with one rank the comparison is with itself, the path runs
        os.environ["RBD_COMM_CHECK"] = "1"
        try:
            out = comm.gatherv(result.vd, [B])
            torch.cuda.synchronize()
            assert torch.equal(out, result.vd)
        finally:
            del os.environ["RBD_COMM_CHECK"]
    finally:
        comm.close()


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["aos", "soa"])
@pytest.mark.parametrize("name", ["atlas_floating", "atlas_fixed", "double_pendulum"])
def test_dynamics_walk_two_states_per_lane_f32(rbd, oracle, models, name, layout, monkeypatch):
    """aba_walk_kernel's packed fp32 form (two states per lane, 128 per workgroup; default from 16 385 states up) forced at a small ragged batch:
    backward error of M v̇ = τ − c against the fp64 oracle's M and c, q̇, and agreement with the one-state-per-lane form."""
    tune(monkeypatch, walk_pair_min_batch="1")
    model = models[name]
    B = 300
    state, q, v, tau, fe = make(rbd, model, B, "f32", layout, 77)
    result = rbd.DynamicsResult(model, B, dtype=torch.float32, layout=layout)
    rbd.dynamics_(result, state, dev(tau, state), dev(fe, state), algorithm="aba_walk")
    assert rbd.sync(state) == 0
    assert "two fp32 states per lane" in rbd.last_kernel(state)
    got, qd = host(result.vd, state), host(result.qd, state)
    assert np.isfinite(got).all()
    M, c = oracle.mass_matrix(model, q), oracle.dynamics_bias(model, q, v, fe)
    Ms = np.tril(M) + np.transpose(np.tril(M, -1), (0, 2, 1))
    res = np.einsum("bij,bj->bi", Ms, got) - (tau - c)
    eta = np.linalg.norm(res, axis=1) / (np.linalg.norm(Ms, axis=(1, 2)) * np.linalg.norm(got, axis=1) + np.linalg.norm(tau - c, axis=1))
    assert eta.max() <= 2e-5, eta.max()
    _, qd_ref = oracle.dynamics(model, q, v, tau, fe, want_qdot=True)
    assert np.abs(qd - qd_ref).max() <= 1e-5 * max(1.0, np.abs(qd_ref).max())
    tune(monkeypatch, walk_pair_min_batch=str(1 << 40))
    state1, *_ = make(rbd, model, B, "f32", layout, 77)
    r1 = rbd.DynamicsResult(model, B, dtype=torch.float32, layout=layout)
    rbd.dynamics_(r1, state1, dev(tau, state1), dev(fe, state1), algorithm="aba_walk")
    assert "two fp32" not in rbd.last_kernel(state1)
    ref = oracle.dynamics(model, q, v, tau, fe)
    assert np.abs(host(r1.vd, state1) - got).max() <= 2e-3 * max(1.0, np.abs(ref).max())


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,B", [("f64", 16384), ("f32", 65536)])
def test_dynamics_walk_full_size(rbd, oracle, models, dtype, B):
    """The walk mapping at the batch sizes it is chosen for: the whole batch against the oracle (fp64: 1e-10; fp32: backward error on every state)."""
    model = models["atlas_floating"]
    state, q, v, tau, fe = make(rbd, model, B, dtype, "aos", 88)
    result = rbd.DynamicsResult(model, B, dtype=TD[dtype])
    rbd.dynamics_(result, state, dev(tau, state), dev(fe, state), algorithm="aba_walk")
    assert rbd.sync(state) == 0
    got = host(result.vd, state)
    if dtype == "f64":
        ref = oracle.dynamics(model, q, v, tau, fe, nthreads=NT)
        assert np.abs(got - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max())
    else:
        M, c = oracle.mass_matrix(model, q, nthreads=NT), oracle.dynamics_bias(model, q, v, fe, nthreads=NT)
        Ms = np.tril(M) + np.transpose(np.tril(M, -1), (0, 2, 1))
        res = np.einsum("bij,bj->bi", Ms, got) - (tau - c)
        eta = np.linalg.norm(res, axis=1) / (np.linalg.norm(Ms, axis=(1, 2)) * np.linalg.norm(got, axis=1) + np.linalg.norm(tau - c, axis=1))
        assert eta.max() <= 2e-5, eta.max()


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["aos", "soa"])
@pytest.mark.parametrize("name", CHAIN_MODELS)
def test_inverse_dynamics_walk_f64(rbd, oracle, models, name, layout):
    """rnea_walk_kernel (one wavefront per track, one lane per state) forced at a small ragged batch: inverse_dynamics! with v̇ and wrenches,
    dynamics_bias!, q̇ untouched paths; and its packed fp32 form."""
    model = models[name]
    B = 70
    state, q, v, tau, fe = make(rbd, model, B, "f64", layout, 91)
    vd = np.random.default_rng(6).standard_normal((B, model.nv))
    out = torch.zeros_like(state.v)
    rbd.inverse_dynamics_(out, state, dev(vd, state), dev(fe, state), mapping="walk")
    ref = oracle.inverse_dynamics(model, q, v, vd, fe)
    assert np.abs(host(out, state) - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max())
    rbd.dynamics_bias_(out, state, mapping="walk")
    ref = oracle.dynamics_bias(model, q, v, None)
    assert np.abs(host(out, state) - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max())


@pytest.mark.gpu
def test_inverse_dynamics_walk_full_size_and_pairs(rbd, oracle, models, monkeypatch):
    """At the sizes where RBD_ALGO_ABA picks it by itself: fp64 16 384 states against the oracle on the whole batch; fp32 65 536 states
    (two states per lane) against the fp64 oracle at fp32 accuracy; the dynamics! → inverse_dynamics! round trip closes (test_mechanism_algorithms.jl:729-740)."""
    model = models["atlas_floating"]
    B = 16384
    state, q, v, tau, fe = make(rbd, model, B, "f64", "aos", 92)
    result = rbd.DynamicsResult(model, B)
    rbd.dynamics_(result, state, dev(tau, state), dev(fe, state))
    assert "aba_walk" in rbd.last_kernel(state)  # (aba_walk_kernel, or aba_walk_spec: the same kernel compiled for the mechanism)
    back = torch.zeros_like(state.v)
    rbd.inverse_dynamics_(back, state, result.vd, dev(fe, state))
    assert float((back - dev(tau, state)).abs().max()) <= 1e-9 * max(1.0, float(np.abs(tau).max()))
    vdh = host(result.vd, state)
    ref = oracle.inverse_dynamics(model, q, v, vdh, fe, nthreads=NT)
    # τ = M v̇ + c with |v̇| up to ~1e3 here: the rounding of the sum scales with its largest term, not with |τ| ≤ 1
    assert np.abs(host(back, state) - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max(), np.abs(vdh).max())
    B = 65536
    state, q, v, tau, fe = make(rbd, model, B, "f32", "aos", 93)
    vd = np.random.default_rng(7).standard_normal((B, model.nv))
    out = torch.zeros_like(state.v)
    rbd.inverse_dynamics_(out, state, dev(vd, state), dev(fe, state))
    cast = lambda a: a.astype(np.float32).astype(np.float64)
    ref = oracle.inverse_dynamics(model, cast(q), cast(v), cast(vd), cast(fe), nthreads=NT)
    assert np.abs(host(out, state) - ref).max() <= 2e-4 * max(1.0, np.abs(ref).max())


@pytest.mark.gpu
def test_removed_mappings_are_refused(rbd, models):
    """RBD_ALGO_ABA_CHAINS / _TRACKS / _PIPE (the experiments of rounds 1-2, removed): reserved values, RBD_ERR_UNSUPPORTED — never another kernel in disguise."""
    model = models["atlas_floating"]
    state, q, v, tau, fe = make(rbd, model, 8, "f64", "aos", 5)
    result = rbd.DynamicsResult(model, 8)
    for algorithm in ("aba_chains", "aba_tracks", "aba_pipe"):
        with pytest.raises(rbd._capi.RBDError) as e:
            rbd.dynamics_(result, state, dev(tau, state), algorithm=algorithm)
        assert e.value.status == 3


@pytest.mark.gpu
def test_mass_matrix_solve_large_ragged_batch_whole_square(rbd, oracle, models):
    """The large-batch fp32 route of mass_matrix! + Cholesky (one-lane-per-state CRBA into a staging buffer, tile Cholesky that re-emits M as
    whole block columns through LDS) on a batch that is NOT a multiple of the 16 states of a Cholesky wavefront, with canaries around the
    outputs: the lower triangle against the oracle, the strict upper triangle = its mirror image (what this route documents), nothing written
    outside the B states."""
    model = models["atlas_floating"]
    B, nv = 32768 + 7, model.nv
    state, q, v, tau, _ = make(rbd, model, B, "f32", "aos", 29)
    rhs = dev(tau, state)
    pad = 3
    xbuf = torch.full((B + pad, nv), 777.0, dtype=torch.float32, device="cuda")
    Mbuf = torch.full((B + pad, nv * nv), 777.0, dtype=torch.float32, device="cuda")
    x, Mout = xbuf[:B], Mbuf[:B]
    rbd.mass_matrix_solve_(x, state, rhs, Mout)
    assert rbd.sync(state) == 0
    assert float((xbuf[B:] - 777.0).abs().max()) == 0.0 and float((Mbuf[B:] - 777.0).abs().max()) == 0.0
    Mfull = Mout.reshape(B, nv, nv).transpose(1, 2).double()  # column-major per state
    assert float((Mfull - Mfull.transpose(1, 2)).abs().max()) == 0.0  # the mirror image, bit for bit
    idx = np.concatenate([np.arange(0, B, 16), np.arange(B - 7, B)])  # a state of every wavefront + the ragged tail
    Mo = oracle.mass_matrix(model, q[idx], nthreads=NT)
    Mo = np.tril(Mo) + np.transpose(np.tril(Mo, -1), (0, 2, 1))
    Mg = Mfull[torch.as_tensor(idx, device=Mfull.device)].cpu().numpy()
    assert np.abs(Mg - Mo).max() <= 2e-5 * np.abs(Mo).max()
    xg = x[torch.as_tensor(idx, device=x.device)].double().cpu().numpy()
    r = tau[idx]
    res = np.einsum("bij,bj->bi", Mo, xg) - r
    eta = np.linalg.norm(res, axis=1) / (np.linalg.norm(Mo, axis=(1, 2)) * np.linalg.norm(xg, axis=1) + np.linalg.norm(r, axis=1))
    assert eta.max() <= 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("nv", [7, 14, 22, 30, 38])
def test_tile_cholesky_every_instantiation(rbd, dtype, nv):
    """The MFMA tile kernels (fp32: chol_mfma_kernel, 16 states per wavefront; fp64: chol_mfma64_kernel, 4 states per wavefront) exist in
    one instantiation per number of 4 x 4 tile rows: sizes that land in the 2-, 4-, 6-, 8- and 10-tile forms (identity padding inside the last
    tile), on a batch that fills neither a whole wavefront nor a multiple of one; factor vs LAPACK's, solution vs numpy."""
    import ctypes
    from rigidbodydynamics_jl_amd import _capi
    rng = np.random.default_rng(100 + nv)
    model = rbd.flatten(rbd.rand_tree_mechanism(rng, ["Revolute"] * nv, lambda m, r: m.bodies[-1]))
    assert model.nv == nv
    B = 13
    A = rng.standard_normal((B, nv, nv))
    A = A @ A.transpose(0, 2, 1) + nv * np.eye(nv)
    b = rng.standard_normal((B, nv))
    st = rbd.MechanismState(model, B, dtype=TD[dtype])
    M = torch.as_tensor(np.tril(A).transpose(0, 2, 1).reshape(B, nv * nv).copy(), dtype=TD[dtype]).cuda()
    rhs = torch.as_tensor(b, dtype=TD[dtype]).cuda()
    x, L = torch.zeros_like(rhs), torch.zeros_like(M)
    opts = st._opts()
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    assert _capi.lib().rbd_cholesky_solve(st.ws.handle, B, P(M), P(rhs), P(x), P(L), ctypes.byref(opts)) == 0
    assert rbd.sync(st) == 0
    Lg = np.tril(L.double().cpu().numpy().reshape(B, nv, nv).transpose(0, 2, 1))
    tol = 1e-12 if dtype == "f64" else 2e-5
    assert np.abs(Lg - np.linalg.cholesky(A)).max() <= tol * np.abs(A).max() ** 0.5
    xr = np.linalg.solve(A, b[..., None])[..., 0]
    assert np.abs(x.double().cpu().numpy() - xr).max() <= tol * max(1.0, np.abs(xr).max())
