"""Known-answer vectors (tests/golden/vectors/*.npz, written by tests/golden/make_vectors.py with the INDEPENDENT body-coordinate
Featherstone implementation oracle/featherstone_np.py): the C oracle and the HIP path must both reproduce them.

  CPU: the C restatement of the reference (rbd_oracle) vs the vectors at 1e-11; the vectors regenerate from the generator script
       (neither implementation can drift unnoticed); the model fixtures regenerate from the reference's URDF files when those are
       present (this container; not on the GPU box).
  GPU: every hot-path entry point through the C ABI vs the vectors at the reference's own atol 1e-10
       (test/test_mechanism_algorithms.jl:739)."""
import importlib.util
import json
import os

import numpy as np
import pytest

from conftest import MODELS, ROOT

VEC = os.path.join(ROOT, "tests", "golden", "vectors")
TREE = ["atlas_floating", "atlas_fixed", "valkyrie_floating", "acrobot_urdf", "double_pendulum", "randmech1", "inner_floating"]


def _gen():
    spec = importlib.util.spec_from_file_location("make_vectors", os.path.join(ROOT, "tests", "golden", "make_vectors.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="module")
def vmodels(rbd):
    return _gen().fixture_models(rbd)


def rel(a, b):
    return float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))


@pytest.mark.parametrize("name", TREE)
def test_c_oracle_reproduces_the_independent_vectors(rbd, oracle, vmodels, name):
    m, g = vmodels[name], np.load(os.path.join(VEC, name + ".npz"))
    assert "featherstone_np" in str(g["source"])
    assert rel(oracle.inverse_dynamics(m, g["q"], g["v"], g["vd_in"], g["fext"]), g["tau"]) <= 1e-11
    assert rel(oracle.dynamics_bias(m, g["q"], g["v"], g["fext"]), g["c"]) <= 1e-11
    assert rel(oracle.dynamics_bias(m, g["q"], g["v"]), g["c_nowrench"]) <= 1e-11
    M = oracle.mass_matrix(m, g["q"])
    assert rel(np.tril(M), np.tril(g["M"])) <= 1e-11
    assert rel(oracle.dynamics(m, g["q"], g["v"], g["tau_in"], g["fext"]), g["vdot"]) <= 1e-11   # CRBA + Cholesky route vs textbook ABA
    assert rel(oracle.aba(m, g["q"], g["v"], g["tau_in"], g["fext"]), g["vdot"]) <= 1e-11        # the oracle's own root-frame ABA
    assert rel(oracle.dynamics(m, g["q"], g["v"], g["tau_in"]), g["vdot_nowrench"]) <= 1e-11


def test_four_bar_vectors_are_what_the_oracle_computes(rbd, oracle, vmodels):
    m, g = vmodels["four_bar"], np.load(os.path.join(VEC, "four_bar.npz"))
    r = oracle.dynamics_loops(m, g["q"], g["v"], g["tau_in"], None, stabilize=True)
    for k in ("vdot", "K", "k", "M", "c"):
        assert rel(r[k], g[k]) <= 1e-12, k
    # what no basis choice can change: the constraint K v̇ + k = 0 holds for the stored v̇
    assert np.abs(np.einsum("bcv,bv->bc", g["K"], g["vdot"]) + g["k"]).max() <= 1e-9


def test_vectors_regenerate(rbd):
    assert _gen().main.__call__ is not None
    import sys
    argv, sys.argv = sys.argv, ["make_vectors.py", "--check"]
    try:
        assert _gen().main() <= 1e-13
    finally:
        sys.argv = argv


@pytest.mark.skipif(not os.path.isdir("/root/reference/test/urdf"), reason="the reference checkout exists only in the build container")
def test_model_fixtures_regenerate_from_the_reference_urdfs(rbd):
    """tests/golden/models/*.json are flattened images of the reference's own URDF fixtures: parse them again and compare every array."""
    spec = importlib.util.spec_from_file_location("make_models", os.path.join(ROOT, "tests", "golden", "make_models.py"))
    mm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mm)
    for name, flat in mm.build_all(rbd).items():
        old = rbd.load_flat_model(os.path.join(MODELS, name + ".json"))
        for f in ("parent", "joint_type", "q_offset", "v_offset"):
            assert np.array_equal(getattr(flat, f), getattr(old, f)), (name, f)
        for f in ("joint_axis", "pred_rot", "pred_trans", "inertia_moment", "inertia_cross", "inertia_mass", "gravity"):
            assert np.abs(np.asarray(getattr(flat, f), float) - np.asarray(getattr(old, f), float)).max() <= 1e-15, (name, f)
        assert flat.body_names == old.body_names


# ---------------------------------------------------------------- GPU ----------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["aos", "soa"])
@pytest.mark.parametrize("name", TREE)
def test_gpu_reproduces_the_independent_vectors(rbd, vmodels, name, layout):
    import torch
    m, g = vmodels[name], np.load(os.path.join(VEC, name + ".npz"))
    B = g["q"].shape[0]
    state = rbd.MechanismState(m, B, layout=layout)
    rbd.set_configuration_(state, g["q"])
    rbd.set_velocity_(state, g["v"])

    def dev(a):
        t = torch.as_tensor(np.ascontiguousarray(a))
        return (t.t().contiguous() if layout == "soa" else t).cuda()

    def host(t):
        return (t.t() if layout == "soa" else t).cpu().numpy()

    tol = 1e-10
    out = torch.zeros_like(state.v)
    rbd.inverse_dynamics_(out, state, dev(g["vd_in"]), dev(g["fext"]))
    assert rel(host(out), g["tau"]) <= tol
    result = rbd.DynamicsResult(m, B, layout=layout)
    rbd.dynamics_bias_(result, state, dev(g["fext"]))
    assert rel(host(result.dynamicsbias), g["c"]) <= tol
    rbd.mass_matrix_(result, state)
    Mg = host(result.massmatrix).reshape(B, m.nv, m.nv).transpose(0, 2, 1)
    assert rel(np.tril(Mg), np.tril(g["M"])) <= tol
    algos = ["aba", "aba_lanes", "crba"] + (["aba_walk", "aba_banks"] if name not in ("randmech1", "inner_floating") else [])
    for a in algos:
        rbd.dynamics_(result, state, dev(g["tau_in"]), dev(g["fext"]), algorithm=a)
        assert rel(host(result.vd), g["vdot"]) <= tol, a
    rbd.dynamics_(result, state, dev(g["tau_in"]))
    assert rel(host(result.vd), g["vdot_nowrench"]) <= tol


@pytest.mark.gpu
def test_gpu_reproduces_the_four_bar_vectors(rbd, vmodels):
    import torch
    m, g = vmodels["four_bar"], np.load(os.path.join(VEC, "four_bar.npz"))
    B = g["q"].shape[0]
    state = rbd.MechanismState(m, B)
    rbd.set_configuration_(state, g["q"])
    rbd.set_velocity_(state, g["v"])
    result = rbd.DynamicsResult(m, B)
    rbd.dynamics_(result, state, torch.as_tensor(g["tau_in"]).cuda())
    assert rel(result.vd.cpu().numpy(), g["vdot"]) <= 1e-9
    K = result.constraintjacobian.cpu().numpy().reshape(B, m.nv, m.nc).transpose(0, 2, 1)
    assert rel(K, g["K"]) <= 1e-10 and rel(result.constraintbias.cpu().numpy(), g["k"]) <= 1e-9
