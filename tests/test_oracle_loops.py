"""Oracle pins for the loop-joint branch (constraint_jacobian!, constraint_bias!, dynamics_solve! with loops), on the
four-bar linkage of test/test_simulate.jl:127-227:
  * KKT conditions of dynamics! : M v̇ + c + K'λ = τ and K v̇ = -k                    (src/mechanism_algorithms.jl:828-836)
  * no stabilization: energy conserved to 1e-8 and loop closure to 1e-10 over 1 s of RK4, Δt = 1e-3   (test :203-213)
  * default Baumgarte gains: a 1e-2-scale initial separation decays (test :215-222; here checked over 3 s: < 2e-3·initial)
and on random trees:
  * maximal coordinates (test/test_mechanism_modification.jl:274-318): the same mechanism with every body on its own floating
    joint and every original joint as a loop joint has the same body accelerations as the tree, atol 1e-10
"""
import numpy as np
import pytest


def closure(oracle, model, q):
    H = oracle.transforms(model, q)
    l = model.loops[0]
    pb = np.einsum("bij,j->bi", H[:, l["predecessor"], :9].reshape(-1, 3, 3), l["pred_trans"]) + H[:, l["predecessor"], 9:]
    pa = np.einsum("bij,j->bi", H[:, l["successor"], :9].reshape(-1, 3, 3), l["succ_trans"]) + H[:, l["successor"], 9:]
    return np.linalg.norm(pb - pa, axis=1)


def rk4(oracle, model, q, v, dt, steps, stabilize):
    f = lambda q, v: oracle.dynamics_loops(model, q, v, stabilize=stabilize)["vdot"]  # revolute joints: q̇ = v
    for _ in range(steps):
        k1q, k1v = v, f(q, v)
        k2q, k2v = v + dt / 2 * k1v, f(q + dt / 2 * k1q, v + dt / 2 * k1v)
        k3q, k3v = v + dt / 2 * k2v, f(q + dt / 2 * k2q, v + dt / 2 * k2v)
        k4q, k4v = v + dt * k3v, f(q + dt * k3q, v + dt * k3v)
        q = q + dt / 6 * (k1q + 2 * k2q + 2 * k3q + k4q)
        v = v + dt / 6 * (k1v + 2 * k2v + 2 * k3v + k4v)
    return q, v


def test_four_bar_structure(rbd, models):
    m = models["four_bar"]
    assert (m.n_bodies, m.nq, m.nv, m.n_loops, m.nc) == (3, 3, 3, 1, 5)
    assert m.parent.tolist() == [-1, 0, -1]
    assert (m.loops[0]["predecessor"], m.loops[0]["successor"]) == (1, 2)


@pytest.mark.parametrize("stabilize", [False, True])
def test_kkt_conditions(rbd, oracle, models, stabilize):
    m = models["four_bar"]
    rng = np.random.default_rng(0)
    B = 16
    q = rbd.FOUR_BAR_INITIAL_Q + np.c_[rng.uniform(-0.05, 0.05, B), np.zeros(B), np.zeros(B)]
    v = rbd.FOUR_BAR_INITIAL_V + 0 * q
    tau = rng.random((B, 3))
    r = oracle.dynamics_loops(m, q, v, tau, stabilize=stabilize)
    Ms = np.tril(r["M"]) + np.transpose(np.tril(r["M"], -1), (0, 2, 1))
    res1 = np.einsum("bij,bj->bi", Ms, r["vdot"]) + r["c"] + np.einsum("bcv,bc->bv", r["K"], r["lam"]) - tau
    assert np.abs(res1).max() < 1e-11
    res2 = np.einsum("bcv,bv->bc", r["K"], r["vdot"]) + r["k"]
    # the planar four-bar has rank(K) = 2 of 5 rows; -k lies in range(K) up to the (tiny) out-of-plane components
    assert np.abs(res2).max() < 1e-9
    assert all(np.linalg.matrix_rank(K, tol=1e-9) == 2 for K in r["K"])
    # lambda is the minimum-norm solution: orthogonal to null(K') i.e. in range(K)
    for K, lam in zip(r["K"], r["lam"]):
        P = K @ np.linalg.pinv(K)
        assert np.abs(P @ lam - lam).max() < 1e-9 * max(1.0, np.abs(lam).max())


def test_four_bar_energy_and_closure_no_stabilization(rbd, oracle, models):
    m = models["four_bar"]
    q0 = rbd.FOUR_BAR_INITIAL_Q[None].copy()
    v0 = np.zeros((1, 3))
    assert closure(oracle, m, q0)[0] < 1e-10
    ke0, pe0 = oracle.energy(m, q0, v0)
    q1, v1 = rk4(oracle, m, q0, v0, 1e-3, 1000, stabilize=False)
    ke1, pe1 = oracle.energy(m, q1, v1)
    assert ke1[0] > 1e-2                                  # it moved (test :207)
    assert abs((ke1 + pe1 - ke0 - pe0)[0]) < 1e-8          # test :208-211
    assert closure(oracle, m, q1)[0] < 1e-10               # test :212-213


def test_four_bar_baumgarte_reduces_separation(rbd, oracle, models):
    m = models["four_bar"]
    q0 = rbd.FOUR_BAR_INITIAL_Q[None].copy()
    q0[0, 0] = 1.7
    v0 = rbd.FOUR_BAR_INITIAL_V[None].copy()
    s0 = closure(oracle, m, q0)[0]
    assert s0 > 1e-2                                       # significant separation initially (test :217-218)
    q1, v1 = rk4(oracle, m, q0, v0, 1e-3, 3000, stabilize=True)
    assert closure(oracle, m, q1)[0] < 2e-3 * s0           # critically damped, T_stab = 0.1 s: decays like (1 + t/T) e^{-t/T}


def quat_from_rot(R):
    """Unit quaternion (w, x, y, z) of a rotation matrix (sign free: the joint transform does not depend on it)."""
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        return np.array([0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s])
    i = int(np.argmax(np.diag(R)))
    j, k = (i + 1) % 3, (i + 2) % 3
    s = np.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2
    q = np.zeros(4)
    q[0] = (R[k, j] - R[j, k]) / s
    q[1 + i] = 0.25 * s
    q[1 + j] = (R[j, i] + R[i, j]) / s
    q[1 + k] = (R[k, i] + R[i, k]) / s
    return q


def maximal_state(H, T):
    """Configuration / velocity of the maximal-coordinates mechanism equivalent to a tree state: every body's floating joint takes the
    body's transform to the root and its twist expressed in the body frame (test/test_mechanism_modification.jl:289-298)."""
    B, nb = H.shape[:2]
    q, v = np.zeros((B, 7 * nb)), np.zeros((B, 6 * nb))
    for b in range(B):
        for i in range(nb):
            R, p = H[b, i, :9].reshape(3, 3), H[b, i, 9:]
            q[b, 7 * i:7 * i + 4] = quat_from_rot(R)
            q[b, 7 * i + 4:7 * i + 7] = p
            w, lin = T[b, i, :3], T[b, i, 3:]
            v[b, 6 * i:6 * i + 3] = R.T @ w
            v[b, 6 * i + 3:6 * i + 6] = R.T @ (lin - np.cross(p, w))
    return q, v


MC_JOINTS = ["QuaternionFloating"] + ["Revolute"] * 2 + ["Prismatic"] + ["Fixed"] + ["QuaternionSpherical"] + ["Planar"] + ["SinCosRevolute"]  # as the reference test: every joint type


@pytest.mark.parametrize("seed", [53, 54, 55])
def test_maximal_coordinates_accelerations(rbd, oracle, seed):
    rng = np.random.default_rng(seed)
    tree = rbd.rand_tree_mechanism(rng, MC_JOINTS)
    mt, mc = rbd.flatten(tree), rbd.flatten(rbd.maximal_coordinates(tree))
    assert (mc.n_bodies, mc.nv, mc.n_loops) == (mt.n_bodies, 6 * mt.n_bodies, mt.n_bodies)
    B = 3
    q, v = rbd.rand_configuration(mt, B, rng), rbd.rand_velocity(mt, B, rng)
    vd = oracle.dynamics(mt, q, v)
    H, T, A_tree = oracle.body_kinematics(mt, q, v, vd)
    qm, vm = maximal_state(H, T)
    out = oracle.dynamics_loops(mc, qm, vm)  # default stabilization gains, as in the reference test (the state is consistent)
    Hm, Tm, A_mc = oracle.body_kinematics(mc, qm, vm, out["vdot"])
    assert np.abs(Hm - H).max() < 1e-12 and np.abs(Tm - T).max() < 1e-12  # same state
    assert np.abs(A_mc - A_tree).max() <= 1e-9 * max(1.0, np.abs(A_tree).max())
