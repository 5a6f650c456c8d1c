"""CPU emulation of aba_pipe_kernel (a body-step cut into stages on four wavefronts, 16 states x 4 tracks per wavefront): the kernel's own
__host__ __device__ stage code (csrc/rbd_pipe.hpp), run lane by lane by tests/emu/pipe_emu.hip on the plan records of
rbd_model_track_plan, against the oracle.  No GPU needed: this checks the arithmetic of the stages, the ring-slot bookkeeping between them
(the four wavefronts are run one after the other between two barriers, in both orders, on an LDS image pre-filled with NaN) and the
mailboxes between the tracks.  fp64 tolerance: the reference's own 1e-10 (test/test_mechanism_algorithms.jl:739)."""
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
    so, src = os.path.join(EMU_DIR, "libpipe_emu.so"), os.path.join(EMU_DIR, "pipe_emu.hip")
    deps = [src] + [os.path.join(CSRC, f) for f in ("rbd_pipe.hpp", "rbd_walk.hpp", "rbd_walk_plan.hpp", "rbd_track.hpp", "rbd_device.hpp")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-x", "hip", "--cuda-host-only", "-O2", "-std=c++17", "-fPIC", "-shared", "-I" + CSRC,
                               "-I" + os.path.join(ROOT, "include"), src, "-o", so])
    L = ctypes.CDLL(so)
    L.pipe_emu_dynamics.restype = ctypes.c_int
    return L


def run_emu(emu, rbd, model, q, v, tau, fe, dtype=np.float64, aos=True, want_qdot=True, reverse=0):
    plan = rbd.track_plan(model)
    assert plan is not None
    B = q.shape[0]
    conv = (lambda a: None if a is None else np.ascontiguousarray(a if aos else a.T, dtype=dtype))
    q_, v_, t_, f_ = conv(q), conv(v), conv(tau), conv(fe)
    vd = np.full((B, model.nv) if aos else (model.nv, B), np.nan, dtype)
    qd = np.full((B, model.nq) if aos else (model.nq, B), np.nan, dtype)
    g = np.ascontiguousarray(model.gravity, np.float64)
    info = np.zeros(2, np.int32)
    p = lambda a: None if a is None else a.ctypes.data_as(ctypes.c_void_p)
    st = emu.pipe_emu_dynamics(p(plan["dims"]), p(plan["ri"]), p(plan["rr"]), p(g), 1 if dtype == np.float32 else 0, int(reverse), int(aos), ctypes.c_long(B),
                               model.nq, model.nv, model.n_bodies, p(q_), p(v_), p(t_), p(f_), p(vd), p(qd) if want_qdot else None, p(info))
    assert st == 0, st
    return (vd if aos else vd.T).astype(np.float64), (qd if aos else qd.T).astype(np.float64), info


@pytest.mark.parametrize("reverse", [0, 1])
@pytest.mark.parametrize("aos", [True, False])
@pytest.mark.parametrize("name", ["atlas_floating", "atlas_fixed", "valkyrie_floating", "acrobot_urdf", "double_pendulum"])
def test_pipe_emulation_matches_oracle_f64(emu, rbd, oracle, models, name, aos, reverse):
    model = models[name]
    if rbd.track_plan(model)["general"]:
        pytest.skip("prismatic / fixed / sin-cos joints take the other mappings")
    B = 21  # one full group of 16 states and a ragged one
    q, v, tau, fe = rand_inputs(rbd, model, B, 91, fext=True)
    ref, qd_ref = oracle.dynamics(model, q, v, tau, fe, want_qdot=True)
    got, qd, info = run_emu(emu, rbd, model, q, v, tau, fe, aos=aos, reverse=reverse)
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max())
    assert np.abs(qd - qd_ref).max() <= 1e-13 * max(1.0, np.abs(qd_ref).max())
    assert info[1] <= 160 * 1024
    # no torques, no wrenches, no q̇
    ref = oracle.dynamics(model, q, v)
    got, _, _ = run_emu(emu, rbd, model, q, v, None, None, aos=aos, want_qdot=False, reverse=reverse)
    assert np.abs(got - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max())


def test_pipe_emulation_f32(emu, rbd, oracle, models):
    model = models["atlas_floating"]
    B = 40
    q, v, tau, fe = rand_inputs(rbd, model, B, 92, fext=True)
    got, _, _ = run_emu(emu, rbd, model, q, v, tau, fe, dtype=np.float32)
    back = oracle.inverse_dynamics(model, q, v, got, fe)
    cb = oracle.dynamics_bias(model, q, v, fe)
    assert (np.linalg.norm(back - tau, axis=1) / np.linalg.norm(tau - cb, axis=1)).max() <= 2e-4


def test_pipe_emulation_random_trees(emu, rbd, oracle):
    """Random revolute trees, from chain-like to bushy, with and without a 6-dof root (rand_tree_mechanism, src/mechanism_modification.jl:382-396)."""
    rng = np.random.default_rng(5)
    done = 0
    for trial in range(60):
        floating, bias, n = bool(trial % 2), float(rng.uniform(0, 1)), int(rng.integers(1, 30))
        types = (["QuaternionFloating"] if floating else []) + ["Revolute"] * n

        def selector(mech, r):
            bodies = [b for b in mech.bodies if not (floating and b is mech.bodies[0])]  # the 6-dof joint stays the only child of the world
            return bodies[-1] if r.random() < bias else bodies[r.integers(len(bodies))]

        try:
            model = rbd.flatten(rbd.rand_tree_mechanism(rng, types, selector))
            plan = rbd.track_plan(model)
        except Exception:
            continue  # more children per body than the library takes
        if plan is None or plan["steps"] > 11:
            continue
        assert not plan["general"]
        done += 1
        B = 5
        q, v, tau, fe = rand_inputs(rbd, model, B, 700 + trial, fext=True)
        ref, qd_ref = oracle.dynamics(model, q, v, tau, fe, want_qdot=True)
        got, qd, _ = run_emu(emu, rbd, model, q, v, tau, fe, reverse=trial % 2)
        assert np.abs(got - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max()), trial
        assert np.abs(qd - qd_ref).max() <= 1e-12 * max(1.0, np.abs(qd_ref).max()), trial
    assert done >= 20, done
