import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

MODELS = os.path.join(ROOT, "tests", "golden", "models")
# the kernels compiled per mechanism: the first call WAITS for hiprtc here, so that a test of a compiled kernel never silently runs the interpreting one
# (the default — compile in the background, interpreting kernels meanwhile — has its own tests: test_jit_cpu.py, test_state_kernels.py)
os.environ.setdefault("RBD_JIT_ASYNC", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def rbd():
    import rbd_amd
    if not os.path.exists(rbd_amd._capi.LIB_PATH):  # fresh checkout: build in-tree (hipcc cross-compiles without a GPU)
        import subprocess
        subprocess.check_call(["bash", os.path.join(ROOT, "rigidbodydynamics.jl_amd", "csrc", "build.sh")])
    return rbd_amd


@pytest.fixture(scope="session")
def oracle():
    import oracle as o
    o.build()
    return o


@pytest.fixture(scope="session")
def models(rbd):
    return build_models(rbd)


def build_models(rbd):
    """name -> FlatModel: URDF mechanisms from the committed flat-model fixtures, the rest built in code."""
    m = {name: rbd.load_flat_model(os.path.join(MODELS, name + ".json"))
         for name in ("atlas_floating", "atlas_fixed", "acrobot_urdf", "valkyrie_floating")}
    m["double_pendulum"] = rbd.flatten(rbd.double_pendulum())
    m["quickstart_pendulum"] = rbd.flatten(rbd.quickstart_double_pendulum())
    m["four_bar"] = rbd.flatten(rbd.four_bar_linkage())
    # test/test_mechanism_algorithms.jl:1-11 style random trees: mixed joint types, several roots, inner floating joints
    for seed in (1, 2, 3):
        m[f"randmech{seed}"] = rbd.flatten(rbd.randmech(np.random.default_rng(seed)))
    # a 6-dof joint in the middle of a chain (maximal-coordinates style) and a spherical/planar mix below it
    m["inner_floating"] = rbd.flatten(rbd.rand_tree_mechanism(np.random.default_rng(7), ["Revolute", "Prismatic", "QuaternionFloating", "Revolute", "QuaternionSpherical", "Planar", "QuaternionFloating", "Revolute"]))
    # ... and one whose nv is a multiple of 4 (the Cholesky compiled for the mechanism's sparsity then applies): 6 + 3 + 3 + 1 + 1 + 1 + 3 + 1 + 1 = 20
    m["mixed20"] = rbd.flatten(rbd.rand_tree_mechanism(np.random.default_rng(11), ["QuaternionFloating", "QuaternionSpherical", "Planar", "Revolute", "Revolute", "Prismatic",
                                                                                     "QuaternionSpherical", "Revolute", "SinCosRevolute"]))
    # mechanisms with LIMBS: sibling subtrees of the same shape, which the fp32 kernels compiled per mechanism walk in lockstep (rbd_spec.hpp, rbd_jit.hip:
    # merge_limbs) — every way a limb can hang in a tree
    R, P, F, S = "Revolute", "Prismatic", "Fixed", "SinCosRevolute"
    chain = lambda *names: (lambda f: f(f, list(names)))(lambda f, ns: (ns[0], [f(f, ns[1:])] if len(ns) > 1 else []))
    hand = (R, [(R, [(R, [])]), (P, []), (R, [(R, [])])])  # a branch point inside a limb (two fingers of the same shape around a thumb: pairs are not nested)
    arm = (R, [(R, [(F, [(R, [hand])])])])
    leg = chain(R, R, P, R, S, R)
    # a humanoid: floating pelvis; torso chain first, the legs behind it (restored from the pelvis's slot); arms around a neck
    m["limbs_humanoid"] = rbd.flatten(rbd.tree_mechanism(np.random.default_rng(21), [("QuaternionFloating", [(R, [(R, [arm, chain(R, R), arm])]), leg, leg])]))
    # two limbs as the ONLY children of a chain body on a fixed base (their sum is the hand-off of a chain parent); axis-aligned constants
    m["limbs_only_children"] = rbd.flatten(rbd.tree_mechanism(np.random.default_rng(22), [(R, [(P, [chain(R, S, R), chain(R, S, R)])])], axis_aligned=True))
    # a quadruped (two pairs under one body, the first pair the first children) with a spherical tail and a planar attachment that stay single
    m["limbs_quadruped"] = rbd.flatten(rbd.tree_mechanism(np.random.default_rng(23), [("QuaternionFloating", [chain(R, R, R), chain(R, R, R), chain(R, R, R), chain(R, R, R),
                                                                                                                  ("QuaternionSpherical", [(R, [])]), ("Planar", [])])], axis_aligned=True))
    # three limbs of one shape (a pair and a single), single-body limbs, limbs ending in fixed joints
    m["limbs_three"] = rbd.flatten(rbd.tree_mechanism(np.random.default_rng(24), [(R, [chain(R, P, F), chain(R, P, F), chain(R, P, F), (R, []), (R, []), (P, [])])]))
    return m


LIMBS = ["limbs_humanoid", "limbs_only_children", "limbs_quadruped", "limbs_three"]


def tune(monkeypatch, **kv):
    """RBD_TUNE="key=value,...": the library's developer knobs (thresholds between the lane mappings, routes forced off; csrc/rbd_capi.hip tune()), read when a
    model / workspace is created.  Several calls in one test accumulate."""
    cur = dict(x.split("=", 1) for x in os.environ.get("RBD_TUNE", "").split(",") if "=" in x)
    cur.update({k: str(v) for k, v in kv.items()})
    monkeypatch.setenv("RBD_TUNE", ",".join(f"{k}={v}" for k, v in cur.items()))


def rand_inputs(rbd, model, B, seed, fext=False):
    rng = np.random.default_rng(seed)
    q = rbd.rand_configuration(model, B, rng)
    v = rbd.rand_velocity(model, B, rng)
    tau = rng.random((B, model.nv))
    out = [q, v, tau]
    if fext:
        out.append(rng.random((B, 6 * model.n_bodies)))
    return out
