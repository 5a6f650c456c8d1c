"""CPU emulation of aba_track_kernel: the kernel's own __host__ __device__ step code (csrc/rbd_track.hpp), run lane by lane by
tests/emu/track_emu.hip on the plan records of rbd_model_track_plan, against the oracle.  No GPU needed: this is what checks the
arithmetic of the track mapping (canonical body frames, folded bias accelerations, 6-dof root) and its plan / mailbox
bookkeeping before the GPU parity tests do.  fp64 tolerance: the reference's own 1e-10 (test/test_mechanism_algorithms.jl:739)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, rand_inputs

EMU_DIR = os.path.join(ROOT, "tests", "emu")
CSRC = os.path.join(ROOT, "rigidbodydynamics.jl_amd", "csrc")


@pytest.fixture(scope="session")
def emu():
    so, src = os.path.join(EMU_DIR, "libtrack_emu.so"), os.path.join(EMU_DIR, "track_emu.hip")
    deps = [src] + [os.path.join(CSRC, f) for f in ("rbd_track.hpp", "rbd_device.hpp")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-x", "hip", "--cuda-host-only", "-O2", "-std=c++17", "-fPIC", "-shared", "-I" + CSRC,
                               "-I" + os.path.join(ROOT, "include"), src, "-o", so])
    L = ctypes.CDLL(so)
    L.track_emu_dynamics.restype = ctypes.c_int
    return L


def run_emu(emu, rbd, model, q, v, tau, fe, dtype=np.float64, aos=True, want_qdot=True, nw=1):
    plan = rbd.track_plan(model)
    assert plan is not None
    B = q.shape[0]
    conv = (lambda a: None if a is None else np.ascontiguousarray(a if aos else a.T, dtype=dtype))
    q_, v_, t_, f_ = conv(q), conv(v), conv(tau), conv(fe)
    vd = np.full((B, model.nv) if aos else (model.nv, B), np.nan, dtype)
    qd = np.full((B, model.nq) if aos else (model.nq, B), np.nan, dtype)
    g = np.ascontiguousarray(model.gravity, np.float64)
    p = lambda a: None if a is None else a.ctypes.data_as(ctypes.c_void_p)
    st = emu.track_emu_dynamics(p(plan["dims"]), p(plan["ri"]), p(plan["rr"]), p(g), int(dtype == np.float32), int(nw), int(aos), ctypes.c_long(B),
                                model.nq, model.nv, model.n_bodies, p(q_), p(v_), p(t_), p(f_), p(vd), p(qd) if want_qdot else None)
    assert st == 0
    return (vd if aos else vd.T).astype(np.float64), (qd if aos else qd.T).astype(np.float64)


@pytest.mark.parametrize("nw", [1, 4])
@pytest.mark.parametrize("aos", [True, False])
@pytest.mark.parametrize("name", ["atlas_floating", "atlas_fixed", "valkyrie_floating", "double_pendulum", "acrobot_urdf"])
def test_track_emulation_matches_oracle_f64(emu, rbd, oracle, models, name, aos, nw):
    model = models[name]
    B = 37  # ragged against 16 / 32 / 64 states per wavefront
    q, v, tau, fe = rand_inputs(rbd, model, B, 71, fext=True)
    ref, qd_ref = oracle.dynamics(model, q, v, tau, fe, want_qdot=True)
    got, qd = run_emu(emu, rbd, model, q, v, tau, fe, aos=aos, nw=nw)
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max())
    assert np.abs(qd - qd_ref).max() <= 1e-13 * max(1.0, np.abs(qd_ref).max())
    # defaults: no torques, no wrenches, no velocities (the M⁻¹ solve runs the pass like this with g = 0)
    ref = oracle.dynamics(model, q, v)
    got, _ = run_emu(emu, rbd, model, q, v, None, None, aos=aos, want_qdot=False, nw=nw)
    assert np.abs(got - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max())


def test_track_emulation_random_trees(emu, rbd, oracle):
    """Random revolute / prismatic / fixed / sin-cos revolute trees with and without a floating root (the reference's randomized-tree style)."""
    from test_chain_plan import random_tree
    rng = np.random.default_rng(19)
    done = 0
    for trial in range(40):
        mech = random_tree(rbd, rng, int(rng.integers(1, 34)), bool(trial % 2), float(rng.uniform(0, 1)))
        model = rbd.flatten(mech)
        plan = rbd.track_plan(model)
        assert plan is not None
        done += 1
        B = 9
        q, v, tau, fe = rand_inputs(rbd, model, B, 100 + trial, fext=True)
        ref = oracle.dynamics(model, q, v, tau, fe)
        got, _ = run_emu(emu, rbd, model, q, v, tau, fe, nw=(1, 4)[(trial // 2) % 2])
        assert np.isfinite(got).all(), trial
        assert np.abs(got - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max()), (trial, plan["tracks"], plan["steps"])
    assert done >= 10


@pytest.mark.parametrize("nw", [1, 4])
def test_track_emulation_f32_backward_error(emu, rbd, oracle, models, nw):
    model = models["atlas_floating"]
    B = 33
    q, v, tau, fe = rand_inputs(rbd, model, B, 72, fext=True)
    q, v, tau, fe = [a.astype(np.float32).astype(np.float64) for a in (q, v, tau, fe)]
    got, _ = run_emu(emu, rbd, model, q, v, tau, fe, dtype=np.float32, nw=nw)
    M = oracle.mass_matrix(model, q)
    M = np.tril(M) + np.transpose(np.tril(M, -1), (0, 2, 1))
    c = oracle.dynamics_bias(model, q, v, fe)
    r = tau - c
    res = np.einsum("bij,bj->bi", M, got) - r
    assert (np.linalg.norm(res, axis=1) / np.linalg.norm(r, axis=1)).max() <= 2e-5


def test_track_plan_invariants(rbd, models):
    for name in ("atlas_floating", "atlas_fixed", "valkyrie_floating", "double_pendulum"):
        model = models[name]
        plan = rbd.track_plan(model)
        tab = plan["table"]
        bodies = sorted(int(x) for x in tab.ravel() if x >= 0)
        assert bodies == list(range(model.n_bodies))  # every body exactly once
        step = {int(b): s for s in range(tab.shape[0]) for b in tab[s] if b >= 0}
        for b in range(model.n_bodies):
            p = int(model.parent[b])
            if p >= 0:
                assert step[p] < step[b]  # parents strictly earlier
    assert rbd.track_plan(models["randmech1"]) is None  # 3-dof joints: outside the mapping
    atlas = rbd.track_plan(models["atlas_floating"])
    assert atlas["tracks"] == 4 and atlas["steps"] == 11  # the critical path of the tree
