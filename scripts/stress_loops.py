"""One-off stress: the loop-joint branch of dynamics! on random mechanisms in maximal coordinates (every tree joint re-expressed as a
loop joint) against the oracle, and against the tree mechanism's body accelerations."""
import os, sys
os.environ.setdefault("RBD_JIT_ASYNC", "0")  # wait for the kernels compiled per mechanism instead of starting on the interpreting ones
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import rbd_amd as rbd, oracle
from test_oracle_loops import maximal_state
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(99)
TYPES = ["Revolute", "Prismatic", "Fixed", "QuaternionSpherical", "SinCosRevolute", "Planar"]
worst = {"vdot_vs_oracle": 0.0, "acc_vs_tree": 0.0}
for trial in range(N):
    n = int(rng.integers(1, 9))
    types = ["QuaternionFloating"] + [str(rng.choice(TYPES)) for _ in range(n)]
    tree = rbd.rand_tree_mechanism(rng, types)
    mt, mc = rbd.flatten(tree), rbd.flatten(rbd.maximal_coordinates(tree))
    B = 7
    q, v = rbd.rand_configuration(mt, B, rng), rbd.rand_velocity(mt, B, rng)
    vd_tree = oracle.dynamics(mt, q, v)
    H, T, A_tree = oracle.body_kinematics(mt, q, v, vd_tree)
    qm, vm = maximal_state(H, T)
    state = rbd.MechanismState(mc, B); res = rbd.DynamicsResult(mc, B)
    rbd.set_configuration_(state, qm); rbd.set_velocity_(state, vm)
    rbd.dynamics_(res, state)
    assert rbd.sync(state) == 0
    got = res.vd.cpu().numpy()
    ref = oracle.dynamics_loops(mc, qm, vm)
    e1 = np.abs(got - ref["vdot"]).max() / max(1.0, np.abs(ref["vdot"]).max())
    _, _, A_mc = oracle.body_kinematics(mc, qm, vm, got)
    e2 = np.abs(A_mc - A_tree).max() / max(1.0, np.abs(A_tree).max())
    worst["vdot_vs_oracle"], worst["acc_vs_tree"] = max(worst["vdot_vs_oracle"], e1), max(worst["acc_vs_tree"], e2)
    assert e1 < 1e-7 and e2 < 1e-7, (trial, types, e1, e2)
print(f"{N} maximal-coordinates mechanisms ok; worst relative errors {worst}")
