"""The kernels compiled per mechanism (csrc/rbd_spec.hpp), run on the HOST one lane at a time (tests/emu/spec_emu.py) against the oracle: the arithmetic of the
generated straight-line code — plan tables, limbs walked in lockstep, folded constants — checked without a GPU.  The GPU tests (test_state_kernels.py) run the
same programs through hiprtc."""
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
