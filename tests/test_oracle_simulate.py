"""Pins for the numpy restatement of `simulate` / MuntheKaasIntegrator (oracle/simulate_np.py):
  * Acrobot: total energy after `simulate(x, 0.1; Δt = 1e-2)` within 1e-3                test/test_simulate.jl:2-13
  * exp/log on SE(3) are inverse, and the rate formula equals a finite difference          (log_with_time_derivative's contract)
  * floating-base mechanism in free fall: energy conserved, quaternion stays on S³ without renormalisation
  * revolute-only mechanisms: the Munthe-Kaas step reduces to classical RK4 on (q, v)
"""
import numpy as np
import pytest

from conftest import rand_inputs


@pytest.fixture(scope="module")
def sim(oracle):
    import simulate_np
    return simulate_np


def test_acrobot_energy(rbd, oracle, models, sim):
    m = models["acrobot_urdf"]
    q, v, _ = rand_inputs(rbd, m, 4, 60)
    ke0, pe0 = oracle.energy(m, q, v)
    ts, q1, v1 = sim.simulate(m, q, v, 0.1, 1e-2)
    # `while t < final_time; t += Δt` (ode_integrators.jl:311-314): ten additions of 0.01 give 0.09999999999999999 < 0.1, so the
    # reference takes an 11th step — the restatement keeps that behaviour
    assert len(ts) == 12
    ke1, pe1 = oracle.energy(m, q1, v1)
    assert np.abs(ke1 + pe1 - ke0 - pe0).max() < 1e-3


def test_se3_exp_log_roundtrip_and_rate(sim):
    rng = np.random.default_rng(0)
    for _ in range(20):
        prot, ptrans = rng.standard_normal(3) * 0.7, rng.standard_normal(3)
        dq, dp = sim.se3_exp(prot, ptrans)
        w, v = rng.standard_normal(3), rng.standard_normal(3)
        psi, qv, psid, qvd = sim.se3_log_with_rate(dq, dp, w, v)
        assert np.allclose(psi, prot, atol=1e-12) and np.allclose(qv, ptrans, atol=1e-12)
        # finite difference of log along the body twist: T(t+h) = T(t) exp(h (w, v))
        h = 1e-6
        eq, ep = sim.se3_exp(h * w, h * v)
        dq2, dp2 = sim.qmul(dq, eq), dp + sim.qrot(dq) @ ep
        psi2, qv2, _, _ = sim.se3_log_with_rate(dq2, dp2, w, v)
        assert np.allclose((psi2 - psi) / h, psid, atol=1e-5) and np.allclose((qv2 - qv) / h, qvd, atol=1e-5)


@pytest.mark.parametrize("name", ["atlas_floating", "randmech1", "inner_floating"])
def test_free_motion_conserves_energy(rbd, oracle, models, sim, name):
    m = models[name]
    q, v, _ = rand_inputs(rbd, m, 2, 61)
    ke0, pe0 = oracle.energy(m, q, v)
    _, q1, v1 = sim.simulate(m, q, v, 0.02, 1e-3)
    ke1, pe1 = oracle.energy(m, q1, v1)
    scale = np.abs(ke0) + np.abs(pe0)
    assert (np.abs(ke1 + pe1 - ke0 - pe0) / scale).max() < 1e-7
    for i in range(m.n_bodies):
        if int(m.joint_type[i]) in (sim.FLOATING, sim.SPHERICAL):
            o = int(m.q_offset[i])
            assert np.abs(np.linalg.norm(q1[:, o:o + 4], axis=1) - 1).max() < 1e-13
        if int(m.joint_type[i]) == sim.SINCOS:
            o = int(m.q_offset[i])
            assert np.abs(np.linalg.norm(q1[:, o:o + 2], axis=1) - 1).max() < 1e-13


def test_revolute_only_is_classical_rk4(rbd, oracle, models, sim):
    m = models["double_pendulum"]
    q, v, tau = rand_inputs(rbd, m, 1, 62)
    f = lambda q, v: oracle.dynamics(m, q, v, tau)
    dt = 1e-2
    k1q, k1v = v, f(q, v)
    k2q, k2v = v + dt / 2 * k1v, f(q + dt / 2 * k1q, v + dt / 2 * k1v)
    k3q, k3v = v + dt / 2 * k2v, f(q + dt / 2 * k2q, v + dt / 2 * k2v)
    k4q, k4v = v + dt * k3v, f(q + dt * k3q, v + dt * k3v)
    qr = q + dt / 6 * (k1q + 2 * k2q + 2 * k3q + k4q)
    vr = v + dt / 6 * (k1v + 2 * k2v + 2 * k3v + k4v)
    qs, vs = sim.step(m, q[0], v[0], dt, tau[0])
    assert np.allclose(qs, qr[0], atol=1e-14) and np.allclose(vs, vr[0], atol=1e-13)


def _hat(w):
    return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0.0]])


def test_se3_exp_is_the_matrix_exponential(sim):
    """test/test_spatial.jl:248-270: exp(ξ) against the exponential of the 4 x 4 matrix [hat(φ_rot) φ_trans; 0 0] (scipy's expm — no formula
    shared with the restatement), log(exp(ξ)) = ξ, for rotation angles from 0 (100 values within 10 eps) to π − eps, and for a pure translation."""
    from scipy.linalg import expm
    rng = np.random.default_rng(74)
    thetas = np.concatenate([np.linspace(0.0, 10 * np.finfo(float).eps, 100), np.linspace(0.0, np.pi - np.finfo(float).eps, 100)])
    for th in thetas:
        u, w = rng.random(3), rng.random(3)
        prot = u / np.linalg.norm(u) * th * 2 * (rng.random() - 0.5)
        ptrans = w / np.linalg.norm(w) * th * 2 * (rng.random() - 0.5)
        dq, dp = sim.se3_exp(prot, ptrans)
        xi = np.zeros((4, 4)); xi[:3, :3] = _hat(prot); xi[:3, 3] = ptrans
        H = expm(xi)
        assert np.allclose(sim.qrot(dq), H[:3, :3], atol=1e-12) and np.allclose(dp, H[:3, 3], atol=1e-12)
        psi, qv, _, _ = sim.se3_log_with_rate(dq, dp, np.zeros(3), np.zeros(3))
        assert np.allclose(psi, prot, atol=1e-9) and np.allclose(qv, ptrans, atol=1e-9)
    ptrans = rng.random(3)
    dq, dp = sim.se3_exp(np.zeros(3), ptrans)
    psi, qv, _, _ = sim.se3_log_with_rate(dq, dp, np.zeros(3), np.zeros(3))
    assert np.allclose(psi, 0) and np.allclose(qv, ptrans)


def test_se3_exp_wraps_beyond_pi(sim):
    """test/test_spatial.jl:272-278: a rotation by θ > π is the rotation by θ mod 2π (as transforms)."""
    rng = np.random.default_rng(75)
    eps = np.finfo(float).eps
    for th in np.concatenate([np.linspace(np.pi - 10 * eps, np.pi + 10 * eps, 100), np.linspace(np.pi, 6 * np.pi, 100)]):
        w = rng.random(3); w /= np.linalg.norm(w)
        q1, p1 = sim.se3_exp(w * th, np.zeros(3))
        q2, p2 = sim.se3_exp(w * np.mod(th, 2 * np.pi), np.zeros(3))
        assert np.allclose(sim.qrot(q1), sim.qrot(q2), atol=1e-9) and np.allclose(p1, p2, atol=1e-12)


def test_rotation_vector_rate_is_the_derivative_of_the_rotation(sim):
    """test/test_spatial.jl:13-33 (Bortz equation): with φ̇ = rotation_vector_rate(φ, ω), d/dt R(φ) = R hat(ω) — central differences stand in
    for the reference's dual numbers; at φ = 0 the rate is ω itself."""
    rng = np.random.default_rng(62)
    for phi in (rng.random(3), 2.5 * rng.random(3), np.zeros(3)):
        w = rng.random(3)
        phid = sim.rotation_vector_rate(phi, w)
        if np.linalg.norm(phi) == 0:
            assert np.allclose(phid, w)
            continue
        R = sim.qrot(sim.quat_from_rotvec(phi))
        h = 1e-6
        Rd = (sim.qrot(sim.quat_from_rotvec(phi + h * phid)) - sim.qrot(sim.quat_from_rotvec(phi - h * phid))) / (2 * h)
        assert np.allclose(Rd, R @ _hat(w), atol=1e-8)


@pytest.mark.parametrize("name", ["randmech1", "randmech2", "randmech3", "inner_floating", "atlas_floating"])
def test_local_and_global_coordinates_are_consistent(rbd, oracle, models, sim, name):
    """test/test_mechanism_algorithms.jl:800-843 for every joint type at once: global_coordinates(q0, 0) = q0 (Duindam, def. 2.9), and with
    q = global_coordinates(q0, ϕ) and ϕ̇ = the rate local_coordinates! returns for (q, v), moving ϕ along ϕ̇ moves q along
    q̇ = velocity_to_configuration_derivative(q, v) (central differences where the reference uses dual numbers; q̇ from the C oracle)."""
    m = models[name]
    rng = np.random.default_rng(44)
    q0 = rbd.rand_configuration(m, 1, rng)[0]
    assert np.allclose(sim.global_coordinates(m, q0, np.zeros(m.nv)), q0, atol=1e-15)
    for _ in range(5):
        phi = 0.4 * rng.standard_normal(m.nv)
        q = sim.global_coordinates(m, q0, phi)
        v = rng.standard_normal(m.nv)
        phid = sim.local_rate(m, q0, q, v)
        h = 1e-6
        qd_fd = (sim.global_coordinates(m, q0, phi + h * phid) - sim.global_coordinates(m, q0, phi - h * phid)) / (2 * h)
        _, qd = oracle.dynamics(m, q[None], v[None], want_qdot=True)
        assert np.allclose(qd_fd, qd[0], atol=2e-8), np.abs(qd_fd - qd[0]).max()


@pytest.mark.parametrize("name", ["atlas_floating", "atlas_fixed", "valkyrie_floating"])
def test_batched_step_equals_the_one_state_step(rbd, oracle, models, sim, name):
    """simulate_np.step_batch (what bench.py and the GPU tests compare whole batches with) restates `step` with arrays over the batch: same numbers, including
    a state that starts at rest (zero rotation in the SE(3) log / exp: the small-angle branches)."""
    m = models[name]
    q, v, tau = rand_inputs(rbd, m, 6, 61)
    v *= 0.3
    v[0] = 0.0
    assert sim.batchable(m)
    q1, v1 = sim.simulate_batch(m, q, v, 2, 1e-3, tau)
    _, q2, v2 = sim.simulate(m, q, v, 2e-3 - 1e-12, 1e-3, tau=tau)
    assert np.abs(q1 - q2).max() < 1e-13 and np.abs(v1 - v2).max() < 1e-9 * max(1.0, np.abs(v2).max())
