"""The kernels compiled per mechanism (csrc/rbd_spec.hpp), run on the HOST one lane at a time (tests/emu/spec_emu.py) against the oracle: the arithmetic of the
generated straight-line code — plan tables, limbs walked in lockstep, folded constants — checked without a GPU.  The GPU tests (test_state_kernels.py) run the
same programs through hiprtc."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))
import spec_emu  # noqa: E402
from conftest import LIMBS  # noqa: E402

pytestmark = pytest.mark.skipif(not spec_emu.available(), reason="no clang++ to build the host emulation with")

IN_SCOPE = ["atlas_floating", "atlas_fixed", "valkyrie_floating", "double_pendulum", "acrobot_urdf"]
EVERY_JOINT_TYPE = ["randmech1", "randmech2", "randmech3", "inner_floating", "mixed20"]
EXPECTED_PAIRS = {"atlas_floating": 13, "atlas_fixed": 13, "limbs_humanoid": 6 + 10, "limbs_only_children": 3, "limbs_quadruped": 6, "limbs_three": 3 + 1,
                  "double_pendulum": 0, "acrobot_urdf": 0}


def backward_error(oracle, model, q, v, tau, fe, vd):
    M = oracle.mass_matrix(model, q)
    Ms = np.tril(M) + np.transpose(np.tril(M, -1), (0, 2, 1))
    c = oracle.dynamics_bias(model, q, v, fe)
    rhs = (tau if tau is not None else 0.0) - c
    res = np.einsum("bij,bj->bi", Ms, vd.astype(np.float64)) - rhs
    return np.linalg.norm(res, axis=1) / (np.linalg.norm(Ms, axis=(1, 2)) * np.linalg.norm(vd, axis=1) + np.linalg.norm(rhs, axis=1))


@pytest.mark.parametrize("name", IN_SCOPE + EVERY_JOINT_TYPE + LIMBS)
def test_emulated_aba_f32(rbd, oracle, models, name):
    model = models[name]
    src = rbd.jit_source(model, torch.float32, "dynamics")
    if src is None:
        pytest.skip("outside the compiled kernels' scope")
    if name in EXPECTED_PAIRS:  # the limbs the plan walks in lockstep
        assert int(re.search(r"NPAIR = (\d+)", src).group(1)) == EXPECTED_PAIRS[name]
    lib = spec_emu.build(src, "ABA")
    B = 70  # two wavefronts, the second partly filled
    rng = np.random.default_rng(5)
    q = rbd.rand_configuration(model, B, rng)
    v = rbd.rand_velocity(model, B, rng)
    tau = rng.random((B, model.nv))
    fe = rng.random((B, 6 * model.n_bodies))
    vd, qd = spec_emu.aba_f32(lib, model, q, v, tau, fe, want_qdot=True)
    assert np.isfinite(vd).all()
    assert backward_error(oracle, model, q, v, tau, fe, vd).max() <= 2e-6
    _, qd_ref = oracle.dynamics(model, q, v, tau, fe, want_qdot=True)
    assert np.abs(qd - qd_ref).max() <= 2e-6 * max(1.0, np.abs(qd_ref).max())
    vd = spec_emu.aba_f32(lib, model, q, v, tau, None)
    assert backward_error(oracle, model, q, v, tau, None, vd).max() <= 2e-6


@pytest.mark.parametrize("name", IN_SCOPE[:2] + EVERY_JOINT_TYPE)
def test_emulated_aba_f64(rbd, oracle, models, name):
    """Round 6: dynamics! in doubles on the lane-per-state program — generated only for the mechanisms no walk kernel takes (3-dof joints, 6-dof joints below the
    world: the reference's own randmech()); the same template as aba_spec_f32, held to the reference's 1e-10 against the oracle, with and without external wrenches."""
    model = models[name]
    src = rbd.jit_source(model, torch.float64, "dynamics")
    if name in IN_SCOPE:
        assert src is None  # (trees of 1-dof joints below a floating base: the walk kernels are ahead in fp64)
        return
    assert src is not None and "aba_spec_f64" in src and "aba_spec_nofext_f64" in src and "NPAIR = 0" in src
    assert "aba_spec_gst_f64" in src and "aba_spec_gst_nofext_f64" in src  # ... and the one with its spare rows in the HBM stash (two wavefronts per CU)
    lib = spec_emu.build(src, "ABA_F64")
    B = 70
    rng = np.random.default_rng(6)
    q, v, tau = rbd.rand_configuration(model, B, rng), rbd.rand_velocity(model, B, rng), rng.random((B, model.nv))
    fe = rng.random((B, 6 * model.n_bodies))
    ref, qd_ref = oracle.dynamics(model, q, v, tau, fe, want_qdot=True)
    ref0 = oracle.dynamics(model, q, v, tau)
    for stash in (False, True):
        vd, qd = spec_emu.aba_f64(lib, model, q, v, tau, fe, want_qdot=True, stash=stash)
        assert np.isfinite(vd).all() and np.abs(vd - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max())
        assert np.abs(qd - qd_ref).max() <= 1e-13 * max(1.0, np.abs(qd_ref).max())
        vd = spec_emu.aba_f64(lib, model, q, v, tau, None, stash=stash)
        assert np.abs(vd - ref0).max() <= 1e-10 * max(1.0, np.abs(ref0).max())


@pytest.mark.parametrize("dtype", ["f32", "f64"])
@pytest.mark.parametrize("name", IN_SCOPE + EVERY_JOINT_TYPE + LIMBS)
def test_emulated_rnea(rbd, oracle, models, name, dtype):
    """inverse_dynamics! with v̇, wrenches on every body and the per-body outputs; dynamics_bias! (v̇ = 0, no wrenches).  fp32: the limbs in lockstep."""
    model = models[name]
    src = rbd.jit_source(model, torch.float32 if dtype == "f32" else torch.float64, "inverse_dynamics")
    if src is None:
        pytest.skip("outside the compiled kernels' scope")
    npair = int(re.search(r"NPAIR = (\d+)", src).group(1))
    assert npair == (EXPECTED_PAIRS.get(name, npair) if dtype == "f32" else 0)  # fp64 has no packed arithmetic: every body on its own
    lib = spec_emu.build(src, "RNEA_F32" if dtype == "f32" else "RNEA_F64")
    np_t, tol = (np.float32, 3e-5) if dtype == "f32" else (np.float64, 1e-10)
    B = 70
    rng = np.random.default_rng(9)
    q = rbd.rand_configuration(model, B, rng)
    v = rbd.rand_velocity(model, B, rng)
    vd = rng.standard_normal((B, model.nv))
    fe = rng.random((B, 6 * model.n_bodies))
    tau, acc, jw = spec_emu.rnea(lib, model, q, v, vd, fe, dtype=np_t, want_bodies=True)
    ref, jw_ref, acc_ref = oracle.inverse_dynamics_bodies(model, q, v, vd, fe)
    for got, want in ((tau, ref), (acc, acc_ref.reshape(B, -1)), (jw, jw_ref.reshape(B, -1))):
        assert np.isfinite(got).all()
        assert np.abs(got - want).max() <= tol * max(1.0, np.abs(want).max())
    tau = spec_emu.rnea(lib, model, q, v, None, None, dtype=np_t)
    ref = oracle.dynamics_bias(model, q, v, None)
    assert np.abs(tau - ref).max() <= tol * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("dtype", ["f32", "f64"])
@pytest.mark.parametrize("name", IN_SCOPE + EVERY_JOINT_TYPE + LIMBS)
def test_emulated_mass_matrix(rbd, oracle, models, name, dtype):
    """mass_matrix!: the lower triangle with its structural zeros; fp32 with nv a multiple of 4: also the permuted staging triangle the tile Cholesky reads."""
    model = models[name]
    src = rbd.jit_source(model, torch.float32 if dtype == "f32" else torch.float64, "mass_matrix")
    if src is None:
        pytest.skip("outside the compiled kernels' scope")
    lib = spec_emu.build(src, "MASS_F32" if dtype == "f32" else "MASS_F64")
    np_t, tol = (np.float32, 2e-6) if dtype == "f32" else (np.float64, 1e-10)
    B, nv = 70, model.nv
    q = rbd.rand_configuration(model, B, np.random.default_rng(13))
    Mr = oracle.mass_matrix(model, q)  # [b][row][col]: lower triangle valid
    il = np.tril_indices(nv)
    got = spec_emu.crba(lib, model, q, np_t)
    assert np.isfinite(got[:, il[0], il[1]]).all()
    assert np.abs(got[:, il[0], il[1]] - Mr[:, il[0], il[1]]).max() <= tol * max(1.0, np.abs(Mr).max())
    if dtype == "f32" and "RBD_SPEC_CHOL" in src:
        perm = np.array([int(x) for x in re.search(r"constexpr int PERM\[NV\] = \{([^}]*)\}", src).group(1).split(",")])
        got = spec_emu.crba(lib, model, q, np_t, permuted=True)
        pr, pc = perm[il[0]], perm[il[1]]
        vals = got[:, np.maximum(pr, pc), np.minimum(pr, pc)]
        assert np.isfinite(vals).all()
        assert np.abs(vals - Mr[:, il[0], il[1]]).max() <= tol * max(1.0, np.abs(Mr).max())


@pytest.mark.parametrize("dtype", ["f32", "f64"])
@pytest.mark.parametrize("name", IN_SCOPE + EVERY_JOINT_TYPE + ["limbs_humanoid"])
def test_emulated_kinematics_byproducts(rbd, oracle, models, name, dtype):
    """Round 6: momentum_matrix! / center_of_mass / energies, geometric_jacobian! of random paths, momentum / momentum_rate_bias — the three kernels of the
    `kinematics` program (csrc/rbd_spec.hpp kin_spec<T, WHAT>) on the host against the oracle; fp64 at the reference's 1e-12
    (test/test_mechanism_algorithms.jl:527-545)."""
    model = models[name]
    src = rbd.jit_source(model, torch.float32 if dtype == "f32" else torch.float64, "kinematics")
    if src is None:
        pytest.skip("outside the compiled kernels' scope")
    assert all(k + "_spec_" + dtype in src for k in ("kin", "jac", "mom", "energy", "com")) and "NPAIR = 0" in src  # (every body walked on its own)
    lib = spec_emu.build(src, "KIN_F32" if dtype == "f32" else "KIN_F64")
    np_t, tol = (np.float32, 3e-5) if dtype == "f32" else (np.float64, 1e-12)
    B = 70
    rng = np.random.default_rng(23)
    q, v = rbd.rand_configuration(model, B, rng), rbd.rand_velocity(model, B, rng)
    q, v = q.astype(np_t).astype(np.float64), v.astype(np_t).astype(np.float64)
    # bodies in depth-first order (the order of the ENTER ops = the slots of csrc/rbd_capi.hip): children in the reference's order, first child first
    kids = {i: [] for i in range(-1, model.n_bodies)}
    for i, p_ in enumerate(model.parent):
        kids[int(p_)].append(i)
    order, stack = [], list(reversed(kids[-1]))
    while stack:
        i = stack.pop()
        order.append(i)
        stack.extend(reversed(kids[i]))
    slot = {b: s for s, b in enumerate(order)}
    rel = lambda got, ref: np.abs(got - ref).max() / max(1.0, np.abs(ref).max())
    A_ref, _, com_ref = oracle.momentum_matrix(model, q, v)
    ke_ref, pe_ref = oracle.energy(model, q, v)
    h_ref, hb_ref = oracle.momentum(model, q, v)
    for trial in range(3):
        base, body = (int(x) for x in rng.choice(np.arange(-1, model.n_bodies), 2, replace=False))
        jp = jm = 0
        a, b = base, body
        while a != b:  # TreePath(base, body): src/graphs/tree_path.jl:41-63
            if a > b:
                jm |= 1 << slot[a]; a = int(model.parent[a])
            else:
                jp |= 1 << slot[b]; b = int(model.parent[b])
        A, com, en, J, mom = spec_emu.kin(lib, model, q, v, np_t, path=(jp, jm))
        J_ref, _ = oracle.geometric_jacobian(model, q, base, body)
        assert np.isfinite(J).all() and rel(J, J_ref) <= tol
    assert np.isfinite(A).all() and rel(A, A_ref) <= tol
    assert rel(com, com_ref) <= tol and rel(en, np.stack([ke_ref, pe_ref], axis=1)) <= 10 * tol
    # ... and the centre of mass from the walk of its own (what center_of_mass(state) alone launches)
    lib.emu_kin(ctypes.c_long(B), spec_emu._p(np.ascontiguousarray(q.T, dtype=np_t)), None, None, spec_emu._p(com2 := np.full((3, B), np.nan, dtype=np_t)), None, None,
                ctypes.c_ulonglong(0), ctypes.c_ulonglong(0), None)
    assert rel(com2.T, com_ref) <= tol
    assert rel(mom, np.concatenate([h_ref, hb_ref], axis=1)) <= 100 * tol
