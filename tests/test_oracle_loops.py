"""Oracle pins for the loop-joint branch (constraint_jacobian!, constraint_bias!, dynamics_solve! with loops), on the
four-bar linkage of test/test_simulate.jl:127-227:
  * KKT conditions of dynamics! : M v̇ + c + K'λ = τ and K v̇ = -k                    (src/mechanism_algorithms.jl:828-836)
  * no stabilization: energy conserved to 1e-8 and loop closure to 1e-10 over 1 s of RK4, Δt = 1e-3   (test :203-213)
  * default Baumgarte gains: a 1e-2-scale initial separation decays (test :215-222; here checked over 3 s: < 2e-3·initial)
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
