"""CPU emulation of aba_walk_kernel (one wavefront per track, one lane per state): the kernel's own __host__ __device__ step code
(csrc/rbd_walk.hpp), run lane by lane by tests/emu/walk_emu.hip on the plan records of rbd_model_track_plan, against the oracle.  No GPU
needed: this checks the arithmetic (un-composed joints in pass B, re-composed transforms in pass C, canonical body frames, folded bias
accelerations, 6-dof root), the parking-slot / mailbox bookkeeping and — by running the wavefronts one after the other between the kernel's
barriers, in both orders — that every mailbox read is ordered after its write.  fp64 tolerance: the reference's own 1e-10
(test/test_mechanism_algorithms.jl:739)."""
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
    so, src = os.path.join(EMU_DIR, "libwalk_emu.so"), os.path.join(EMU_DIR, "walk_emu.hip")
    deps = [src] + [os.path.join(CSRC, f) for f in ("rbd_walk.hpp", "rbd_walk_plan.hpp", "rbd_track_plan.hpp", "rbd_mk_fuse.hpp", "rbd_device.hpp")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-x", "hip", "--cuda-host-only", "-O2", "-std=c++17", "-fPIC", "-shared", "-I" + CSRC,
                               "-I" + os.path.join(ROOT, "include"), src, "-o", so])
    L = ctypes.CDLL(so)
    L.walk_emu_dynamics.restype = ctypes.c_int
    return L


def run_emu(emu, rbd, model, q, v, tau, fe, dtype=np.float64, aos=True, want_qdot=True, reverse=0, pair=False, rnea=False, reroot=False):
    plan = rbd.reroot_plan(model) if reroot else rbd.track_plan(model)
    assert plan is not None
    B = q.shape[0]
    conv = (lambda a: None if a is None else np.ascontiguousarray(a if aos else a.T, dtype=dtype))
    q_, v_, t_, f_ = conv(q), conv(v), conv(tau), conv(fe)
    vd = np.full((B, model.nv) if aos else (model.nv, B), np.nan, dtype)
    qd = np.full((B, model.nq) if aos else (model.nq, B), np.nan, dtype)
    g = np.ascontiguousarray(model.gravity, np.float64)
    info = np.zeros(2, np.int32)
    p = lambda a: None if a is None else a.ctypes.data_as(ctypes.c_void_p)
    st = emu.walk_emu_dynamics(p(plan["dims"]), p(plan["ri"]), p(plan["rr"]), p(g), (2 if pair else 1) if dtype == np.float32 else 0, int(reverse), int(aos), ctypes.c_long(B),
                               model.nq, model.nv, model.n_bodies, p(q_), p(v_), p(t_), p(f_), p(vd), p(qd) if want_qdot else None, p(info), int(rnea),
                               p(plan["wk"]) if reroot else None, p(plan["chain_i"]) if reroot else None, p(plan["chain_r"]) if reroot else None,
                               p(plan["fxp"]) if reroot else None)
    assert st == 0
    return (vd if aos else vd.T).astype(np.float64), (qd if aos else qd.T).astype(np.float64), info


@pytest.mark.parametrize("reverse", [0, 1])
@pytest.mark.parametrize("aos", [True, False])
@pytest.mark.parametrize("name", ["atlas_floating", "atlas_fixed", "valkyrie_floating", "double_pendulum", "acrobot_urdf"])
def test_walk_emulation_matches_oracle_f64(emu, rbd, oracle, models, name, aos, reverse):
    model = models[name]
    B = 70  # two workgroups, the second partly filled
    q, v, tau, fe = rand_inputs(rbd, model, B, 71, fext=True)
    ref, qd_ref = oracle.dynamics(model, q, v, tau, fe, want_qdot=True)
    got, qd, info = run_emu(emu, rbd, model, q, v, tau, fe, aos=aos, reverse=reverse)
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max())
    assert np.abs(qd - qd_ref).max() <= 1e-13 * max(1.0, np.abs(qd_ref).max())
    # defaults: no torques, no wrenches, no velocities (the M⁻¹ solve runs the pass like this with g = 0)
    ref = oracle.dynamics(model, q, v)
    got, _, _ = run_emu(emu, rbd, model, q, v, None, None, aos=aos, want_qdot=False, reverse=reverse)
    assert np.abs(got - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max())


def test_walk_lds_budget_atlas(emu, rbd, oracle, models):
    """The whole per-workgroup footprint (rows of 64 states, mailboxes, parking slots, plan records) of Atlas fits one CU's LDS in fp64."""
    model = models["atlas_floating"]
    q, v, tau, fe = rand_inputs(rbd, model, 3, 5, fext=True)
    _, _, info = run_emu(emu, rbd, model, q, v, tau, fe)
    assert info[1] <= 160 * 1024, info
    _, _, info32 = run_emu(emu, rbd, model, q, v, tau, fe, dtype=np.float32)
    assert info32[1] <= 80 * 1024 + 2048, info32
    _, _, info2 = run_emu(emu, rbd, model, q, v, tau, fe, dtype=np.float32, pair=True)
    assert info2[1] <= 160 * 1024, info2


@pytest.mark.parametrize("pair", [False, True])
@pytest.mark.parametrize("aos", [True, False])
def test_walk_emulation_f32(emu, rbd, oracle, models, aos, pair):
    """fp32, one state per lane and the packed form (two states per lane, 128 per workgroup: B = 200 leaves the second workgroup ragged in both components)."""
    model = models["atlas_floating"]
    B = 200
    q, v, tau, fe = rand_inputs(rbd, model, B, 72, fext=True)
    got, qd, _ = run_emu(emu, rbd, model, q, v, tau, fe, dtype=np.float32, aos=aos, pair=pair)
    assert np.isfinite(got).all() and np.isfinite(qd).all()
    _, qd_ref = oracle.dynamics(model, q, v, tau, fe, want_qdot=True)
    assert np.abs(qd - qd_ref).max() <= 1e-5 * max(1.0, np.abs(qd_ref).max())
    M, c = oracle.mass_matrix(model, q), oracle.dynamics_bias(model, q, v, fe)
    Ms = np.tril(M) + np.transpose(np.tril(M, -1), (0, 2, 1))
    res = np.einsum("bij,bj->bi", Ms, got) - (tau - c)
    eta = np.linalg.norm(res, axis=1) / (np.linalg.norm(Ms, axis=(1, 2)) * np.linalg.norm(got, axis=1) + np.linalg.norm(tau - c, axis=1))
    assert eta.max() <= 2e-5, eta.max()


def test_walk_emulation_random_trees(emu, rbd, oracle):
    """Random revolute / prismatic / fixed / sin-cos revolute trees with and without a floating root (the reference's randomized-tree style)."""
    from test_chain_plan import random_tree
    rng = np.random.default_rng(23)
    done = 0
    for trial in range(40):
        mech = random_tree(rbd, rng, int(rng.integers(1, 34)), bool(trial % 2), float(rng.uniform(0, 1)))
        model = rbd.flatten(mech)
        plan = rbd.track_plan(model)
        assert plan is not None
        if plan["steps"] > 11:
            continue  # deeper than the accumulation registers hold: the library routes such trees to the other mappings
        done += 1
        B = 5
        q, v, tau, fe = rand_inputs(rbd, model, B, 100 + trial, fext=True)
        ref, qd_ref = oracle.dynamics(model, q, v, tau, fe, want_qdot=True)
        got, qd, _ = run_emu(emu, rbd, model, q, v, tau, fe, reverse=trial % 2)
        assert np.isfinite(got).all(), trial
        assert np.abs(got - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max()), (trial, plan["tracks"], plan["steps"])
        assert np.abs(qd - qd_ref).max() <= 1e-12 * max(1.0, np.abs(qd_ref).max()), trial
    assert done >= 25


@pytest.mark.parametrize("reverse", [0, 1])
@pytest.mark.parametrize("name", ["atlas_floating", "atlas_fixed", "valkyrie_floating", "double_pendulum"])
def test_walk_emulation_inverse_dynamics(emu, rbd, oracle, models, name, reverse):
    """rnea_walk_kernel's step code: inverse_dynamics! with v̇ and external wrenches, and dynamics_bias! (no v̇), against the oracle."""
    model = models[name]
    B = 70
    q, v, tau, fe = rand_inputs(rbd, model, B, 81, fext=True)
    vd = np.random.default_rng(4).standard_normal((B, model.nv))
    ref = oracle.inverse_dynamics(model, q, v, vd, fe)
    got, qd, _ = run_emu(emu, rbd, model, q, v, vd, fe, reverse=reverse, rnea=True)
    assert np.abs(got - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max())
    _, qd_ref = oracle.dynamics(model, q, v, tau, fe, want_qdot=True)
    assert np.abs(qd - qd_ref).max() <= 1e-13 * max(1.0, np.abs(qd_ref).max())
    ref = oracle.dynamics_bias(model, q, v, None)
    got, _, _ = run_emu(emu, rbd, model, q, v, None, None, reverse=reverse, rnea=True, want_qdot=False)
    assert np.abs(got - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max())


def test_walk_emulation_inverse_dynamics_random_trees_and_pairs(emu, rbd, oracle):
    from test_chain_plan import random_tree
    rng = np.random.default_rng(29)
    done = 0
    for trial in range(30):
        mech = random_tree(rbd, rng, int(rng.integers(1, 30)), bool(trial % 2), float(rng.uniform(0, 1)))
        model = rbd.flatten(mech)
        plan = rbd.track_plan(model)
        if plan is None or plan["steps"] > 11 or model.nv == 0:
            continue
        done += 1
        B = 5
        q, v, tau, fe = rand_inputs(rbd, model, B, 300 + trial, fext=True)
        vd = rng.standard_normal((B, model.nv))
        ref = oracle.inverse_dynamics(model, q, v, vd, fe)
        got, _, _ = run_emu(emu, rbd, model, q, v, vd, fe, reverse=trial % 2, rnea=True)
        assert np.abs(got - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max()), trial
        if trial % 3 == 0:  # the packed fp32 form
            got32, _, _ = run_emu(emu, rbd, model, q, v, vd, fe, dtype=np.float32, pair=True, rnea=True)
            assert np.abs(got32 - ref).max() <= 2e-4 * max(1.0, np.abs(ref).max()), trial
    assert done >= 20


@pytest.mark.parametrize("reverse", [0, 1])
@pytest.mark.parametrize("aos", [True, False])
@pytest.mark.parametrize("name", ["atlas_floating", "valkyrie_floating"])
def test_walk_emulation_rerooted_tree(emu, rbd, oracle, models, name, aos, reverse):
    """The walk kernel's step code on the plan of the tree RE-ROOTED at its centre (csrc/rbd_reroot.hpp): Atlas is 11 bodies deep from the pelvis
    and 9 from its middle torso link.  Same v̇ and q̇ as the oracle (whose tree hangs from the pelvis), torques and wrenches on every body."""
    model = models[name]
    plan = rbd.reroot_plan(model)
    assert plan is not None and plan["steps"] < rbd.track_plan(model)["steps"] and plan["chain"] >= 1
    B = 70
    q, v, tau, fe = rand_inputs(rbd, model, B, 171, fext=True)
    ref, qd_ref = oracle.dynamics(model, q, v, tau, fe, want_qdot=True)
    got, qd, _ = run_emu(emu, rbd, model, q, v, tau, fe, aos=aos, reverse=reverse, reroot=True)
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max())
    assert np.abs(qd - qd_ref).max() <= 1e-13 * max(1.0, np.abs(qd_ref).max())
    ref = oracle.dynamics(model, q, v)
    got, _, _ = run_emu(emu, rbd, model, q, v, None, None, aos=aos, want_qdot=False, reverse=reverse, reroot=True)
    assert np.abs(got - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max())


def test_walk_emulation_rerooted_random_floating_trees(emu, rbd, oracle):
    from test_chain_plan import random_tree
    rng = np.random.default_rng(31)
    done = 0
    for trial in range(60):
        mech = random_tree(rbd, rng, int(rng.integers(4, 30)), True, float(rng.uniform(0, 1)))
        model = rbd.flatten(mech)
        plan = rbd.reroot_plan(model)
        if plan is None or plan["steps"] > 11:
            continue
        done += 1
        B = 5
        q, v, tau, fe = rand_inputs(rbd, model, B, 500 + trial, fext=True)
        ref, qd_ref = oracle.dynamics(model, q, v, tau, fe, want_qdot=True)
        got, qd, _ = run_emu(emu, rbd, model, q, v, tau, fe, reverse=trial % 2, reroot=True)
        assert np.abs(got - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max()), (trial, plan["root"], plan["chain"])
        assert np.abs(qd - qd_ref).max() <= 1e-12 * max(1.0, np.abs(qd_ref).max()), trial
        if trial % 3 == 0:
            got32, _, _ = run_emu(emu, rbd, model, q, v, tau, fe, dtype=np.float32, pair=True, reroot=True)
            back = oracle.inverse_dynamics(model, q, v, got32, fe)
            cb = oracle.dynamics_bias(model, q, v, fe)
            assert (np.linalg.norm(back - tau, axis=1) / np.linalg.norm(tau - cb, axis=1)).max() <= 2e-4, trial
    assert done >= 10, done
