"""Pins the CPU oracle (oracle/rbd_oracle.c) to the reference's own tests.  The reference stores no golden
vectors and Julia is unavailable, so the pins are its closed-form and invariant tests:

  test/test_double_pendulum.jl:54-75     closed-form M, C, G of the double pendulum, atol 1e-12
  test/test_double_pendulum.jl:78-99     the same through Acrobot.urdf (incl. a Fixed tree joint)
  test/test_urdf.jl:85-101               RPY golden matrices (ROS tf)
  test/test_mechanism_algorithms.jl:564-572  1/2 v'Mv = kinetic energy
  test/test_mechanism_algorithms.jl:600-614  M = d tau / d vdot
  test/test_mechanism_algorithms.jl:654-675  gravity term = d PE / d q
  test/test_mechanism_algorithms.jl:729-740  dynamics! -> inverse_dynamics round trip, atol 1e-10
  test/test_mechanism_algorithms.jl:742-753  dynamics_bias = inverse_dynamics(vdot = 0)
  test/test_mechanism_algorithms.jl:310-327  geometric Jacobian: J v = relative twist, atol 1e-12
  test/test_mechanism_algorithms.jl:527-545  momentum matrix: A v = sum_b I_b T_b, atol 1e-12
  test/test_mechanism_algorithms.jl:616-652  Coriolis term: dM/dt - 2C is skew-symmetric (finite differences instead of ForwardDiff)
  test/test_mechanism_algorithms.jl:773-797  power flow with external wrenches: tau.v + sum wext.T = d(PE + KE)/dt
  test/test_mechanism_algorithms.jl:707-727  momentum-rate balance with external wrenches (through the
                                              floating-base rows of tau)
"""
import numpy as np
import pytest

from conftest import rand_inputs


def closed_form(q, v, lc1=-0.5, l1=-1.0, m1=1.0, I1=0.333, lc2=-1.0, m2=1.0, I2=1.33, g=-9.81):
    """test/test_double_pendulum.jl:41-65."""
    q1, q2 = q
    v1, v2 = v
    c2, s1, s2, s12 = np.cos(q2), np.sin(q1), np.sin(q2), np.sin(q1 + q2)
    M11 = I1 + I2 + m2 * l1 ** 2 + 2 * m2 * l1 * lc2 * c2
    M12 = I2 + m2 * l1 * lc2 * c2
    M = np.array([[M11, M12], [M12, I2]])
    C = np.array([[-2 * m2 * l1 * lc2 * s2 * v2, -m2 * l1 * lc2 * s2 * v2], [m2 * l1 * lc2 * s2 * v1, 0.0]])
    G = np.array([m1 * g * lc1 * s1 + m2 * g * (l1 * s1 + lc2 * s12), m2 * g * lc2 * s12])
    T1 = 0.5 * I1 * v1 ** 2
    T2 = 0.5 * (m2 * l1 ** 2 + I2 + 2 * m2 * l1 * lc2 * c2) * v1 ** 2 + 0.5 * I2 * v2 ** 2 + (I2 + m2 * l1 * lc2 * c2) * v1 * v2
    return M, C, G, T1 + T2


@pytest.mark.parametrize("name", ["double_pendulum", "acrobot_urdf"])
def test_double_pendulum_closed_form(oracle, models, name):
    model = models[name]
    assert (model.nq, model.nv) == (2, 2)
    rng = np.random.default_rng(6)
    B = 64
    q, v, vd = rng.standard_normal((B, 2)), rng.random((B, 2)), rng.random((B, 2))
    Mo = oracle.mass_matrix(model, q)
    tau = oracle.inverse_dynamics(model, q, v, vd)
    ke, _ = oracle.energy(model, q, v)
    for b in range(B):
        M, C, G, T = closed_form(q[b], v[b])
        assert np.allclose(np.tril(Mo[b]), np.tril(M), atol=1e-12, rtol=0)
        assert np.allclose(tau[b], M @ vd[b] + C @ v[b] + G, atol=1e-12, rtol=0)
        assert abs(ke[b] - T) < 1e-12


def test_quickstart_example_state(oracle, models):
    """BASELINE.json configs[0]: examples/1 double pendulum at q = (0.3, 0.4), v = (1, 2) (example :103-106)."""
    model = models["quickstart_pendulum"]
    q, v = np.array([[0.3, 0.4]]), np.array([[1.0, 2.0]])
    vd = oracle.dynamics(model, q, v)
    M, C, G, _ = closed_form(q[0], v[0], lc1=-0.5, l1=-1.0, I1=0.333, lc2=-0.5, I2=0.333)
    assert np.allclose(vd[0], np.linalg.solve(M, -(C @ v[0] + G)), atol=1e-12)
    assert np.allclose(oracle.aba(model, q, v)[0], vd[0], atol=1e-12)


def test_rpy_golden(rbd):
    """test/test_urdf.jl:85-101 (ROS tf golden matrices)."""
    import xml.etree.ElementTree as ET
    R, p = rbd.parse_pose(ET.fromstring('<origin rpy="1 2 3"/>'))
    assert np.allclose(R, [[0.41198225, -0.83373765, -0.36763046], [-0.05872664, -0.42691762, 0.90238159],
                           [-0.90929743, -0.35017549, -0.2248451]], atol=1e-7)
    assert np.allclose(p, 0)
    R, _ = rbd.parse_pose(ET.fromstring('<origin rpy="0.5 0.1 0.2"/>'))
    assert np.allclose(R, [[0.97517033, -0.12744012, 0.18111281], [0.19767681, 0.86959819, -0.45246312],
                           [-0.09983342, 0.47703041, 0.8731983]], atol=1e-7)


def test_atlas_flattening_checksums(models):
    """joint order (BFS), parents, total mass (fixed children merged)."""
    af = models["atlas_floating"]
    assert (af.n_bodies, af.nq, af.nv) == (31, 37, 36)
    assert abs(af.total_mass() - 175.117964) < 1e-9
    assert af.parent.tolist() == [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 7, 7, 8, 9, 10, 12, 13, 14, 15, 16, 17, 18, 19, 20, 23, 24, 25, 26, 27, 28]
    assert af.joint_names[:4] == ["pelvis_to_world", "back_bkz", "l_leg_hpz", "r_leg_hpz"]
    assert np.bincount(af.levels()).tolist() == [1, 3, 3, 3, 5, 4, 4, 2, 2, 2, 2]
    ax = models["atlas_fixed"]
    assert (ax.n_bodies, ax.nq, ax.nv) == (30, 30, 30)


MODELS = ["atlas_floating", "atlas_fixed", "valkyrie_floating", "double_pendulum", "randmech1", "randmech2", "randmech3", "inner_floating"]


@pytest.mark.parametrize("name", MODELS)
def test_kinetic_energy_is_half_vMv(rbd, oracle, models, name):
    model = models[name]
    q, v, _ = rand_inputs(rbd, model, 8, 1)
    M = oracle.mass_matrix(model, q)
    ke, _ = oracle.energy(model, q, v)
    for b in range(8):
        Ms = np.tril(M[b]) + np.tril(M[b], -1).T
        assert abs(0.5 * v[b] @ Ms @ v[b] - ke[b]) < 1e-10 * max(1.0, abs(ke[b]))


@pytest.mark.parametrize("name", MODELS)
def test_mass_matrix_is_dtau_dvdot_and_bias_is_id0(rbd, oracle, models, name):
    model = models[name]
    B = 3
    q, v, _ = rand_inputs(rbd, model, B, 2)
    M = oracle.mass_matrix(model, q)
    c = oracle.dynamics_bias(model, q, v)
    id0 = oracle.inverse_dynamics(model, q, v, np.zeros((B, model.nv)))
    assert np.allclose(c, id0, atol=1e-12, rtol=1e-12)
    for i in range(model.nv):
        e = np.zeros((B, model.nv))
        e[:, i] = 1.0
        col = oracle.inverse_dynamics(model, q, v, e) - id0
        for b in range(B):
            Ms = np.tril(M[b]) + np.tril(M[b], -1).T
            assert np.allclose(col[b], Ms[:, i], atol=1e-9 * max(1.0, np.abs(Ms).max()))


@pytest.mark.parametrize("name", MODELS)
def test_dynamics_inverse_dynamics_round_trip(rbd, oracle, models, name):
    """dynamics! then inverse_dynamics gives tau back (atol 1e-10 in the reference; scaled by |tau - c| here)."""
    model = models[name]
    B = 16
    q, v, tau, fext = rand_inputs(rbd, model, B, 3, fext=True)
    vd = oracle.dynamics(model, q, v, tau, fext)
    back = oracle.inverse_dynamics(model, q, v, vd, fext)
    c = oracle.dynamics_bias(model, q, v, fext)
    scale = np.abs(tau - c).max()
    assert np.abs(back - tau).max() < 1e-10 * max(1.0, scale)


@pytest.mark.parametrize("name", MODELS)
def test_aba_matches_reference_route(rbd, oracle, models, name):
    """The independent world-frame ABA reproduces CRBA + Cholesky."""
    model = models[name]
    B = 32
    q, v, tau, fext = rand_inputs(rbd, model, B, 4, fext=True)
    a = oracle.dynamics(model, q, v, tau, fext)
    b = oracle.aba(model, q, v, tau, fext)
    assert np.abs(a - b).max() <= 1e-10 * max(1.0, np.abs(a).max())


@pytest.mark.parametrize("name", ["atlas_floating", "atlas_fixed"])
def test_gravity_term_is_dPE_dq(rbd, oracle, models, name):
    """c(q, v=0)·v_dir = d PE / dt along q̇(v_dir): test/test_mechanism_algorithms.jl:654-675 (finite differences)."""
    model = models[name]
    B = 4
    q, v, _ = rand_inputs(rbd, model, B, 5)
    g = oracle.dynamics_bias(model, q, np.zeros_like(v))
    _, qd = oracle.dynamics(model, q, v, want_qdot=True)
    h = 1e-6
    _, pe_p = oracle.energy(model, q + h * qd, v)
    _, pe_m = oracle.energy(model, q - h * qd, v)
    dpe = (pe_p - pe_m) / (2 * h)
    assert np.allclose(np.einsum("bi,bi->b", g, v), dpe, rtol=1e-6, atol=1e-5)


def test_momentum_rate_balance_floating(rbd, oracle, models):
    """With a floating base, the base rows of inverse dynamics are the total momentum rate minus external wrenches
    expressed in the base frame; with vdot from dynamics! and tau_base = 0 the whole-body Newton–Euler balance
    (test/test_mechanism_algorithms.jl:707-727) closes: tau round trip already checks it; here check symmetry and
    positive-definiteness of M instead (Cholesky succeeded) and the power balance d(KE+PE)/dt = tau·v."""
    model = models["atlas_floating"]
    B = 4
    q, v, tau = rand_inputs(rbd, model, B, 7)
    vd, qd = oracle.dynamics(model, q, v, tau, want_qdot=True)
    h = 1e-6
    # normalise the quaternion after stepping (q̇ is tangent to S³ only to first order)
    def step(s):
        qq = q + s * qd
        qq[:, :4] /= np.linalg.norm(qq[:, :4], axis=1, keepdims=True)
        return qq
    kp, pp = oracle.energy(model, step(h), v + h * vd)
    km, pm = oracle.energy(model, step(-h), v - h * vd)
    dE = (kp + pp - km - pm) / (2 * h)
    assert np.allclose(dE, np.einsum("bi,bi->b", tau, v), rtol=1e-5, atol=1e-4)


def test_f32_oracle_tracks_f64(rbd, oracle, models):
    """fp32 build of the same restatement: backward error of v̇ small (forward error is cond(M)-limited:
    SURVEY.md App. B, precedent atol 1e-3 test/test_mechanism_modification.jl:339)."""
    model = models["atlas_floating"]
    B = 16
    q, v, tau = rand_inputs(rbd, model, B, 8)
    vd32 = oracle.aba(model, q, v, tau, dtype=np.float32).astype(np.float64)
    back = oracle.inverse_dynamics(model, q, v, vd32)
    c = oracle.dynamics_bias(model, q, v)
    rel = np.linalg.norm(back - tau, axis=1) / np.linalg.norm(tau - c, axis=1)
    assert rel.max() < 1e-4


@pytest.mark.parametrize("name", MODELS)
def test_momentum_matrix_times_v_is_total_momentum(rbd, oracle, models, name):
    """test/test_mechanism_algorithms.jl:527-545 (A·v against Σ_b I_b T_b) plus center_of_mass against PE = -m g·com."""
    model = models[name]
    q, v, _ = rand_inputs(rbd, model, 8, 5)
    A, h, com = oracle.momentum_matrix(model, q, v)
    _, pe = oracle.energy(model, q, v)
    g = np.asarray(model.gravity, float)
    mass = float(np.sum(model.inertia_mass))
    for b in range(8):
        assert np.abs(A[b] @ v[b] - h[b]).max() <= 1e-12 * max(1.0, np.abs(h[b]).max())
        assert abs(-mass * g @ com[b] - pe[b]) <= 1e-12 * max(1.0, abs(pe[b]))


@pytest.mark.parametrize("name", MODELS)
def test_geometric_jacobian_times_v_is_relative_twist(rbd, oracle, models, name):
    """test/test_mechanism_algorithms.jl:310-327: random (base, body) pairs, the root body included."""
    model = models[name]
    q, v, _ = rand_inputs(rbd, model, 4, 6)
    rng = np.random.default_rng(7)
    for _ in range(10):
        base, body = rng.choice(np.arange(-1, model.n_bodies), 2, replace=False)
        J, t = oracle.geometric_jacobian(model, q, base, body, v)
        for b in range(4):
            assert np.abs(J[b] @ v[b] - t[b]).max() <= 1e-12 * max(1.0, np.abs(t[b]).max())
    J, t = oracle.geometric_jacobian(model, q, 0, 0, v)
    assert np.abs(J).max() == 0.0


@pytest.mark.parametrize("name", ["randmech1", "randmech2", "randmech3", "atlas_floating", "inner_floating"])
def test_power_flow_with_external_wrenches(rbd, oracle, models, name):
    """test/test_mechanism_algorithms.jl:773-797: the power of the joint torques and of the external wrenches equals the rate of the
    total energy along (q̇, v̇) of dynamics!.  The reference differentiates with dual numbers; here a central difference (h = 1e-6,
    quaternion blocks drift off the unit sphere only by O(h²))."""
    model = models[name]
    B = 4
    q, v, tau, fe = rand_inputs(rbd, model, B, 21, fext=True)
    vd, qd = oracle.dynamics(model, q, v, tau, fe, want_qdot=True)
    power = np.einsum("bi,bi->b", tau, v)
    for body in range(model.n_bodies):
        _, twist = oracle.geometric_jacobian(model, q, -1, body, v)  # twist_wrt_world of the body, root frame
        power += np.einsum("bk,bk->b", fe[:, 6 * body:6 * body + 6], twist)
    h = 1e-6
    kp, pp = oracle.energy(model, q + h * qd, v + h * vd)
    km, pm = oracle.energy(model, q - h * qd, v - h * vd)
    dE = (kp + pp - km - pm) / (2 * h)
    assert np.allclose(dE, power, rtol=1e-6, atol=1e-5 * max(1.0, np.abs(power).max()))


def test_coriolis_matrix_skew_symmetry(rbd, oracle):
    """test/test_mechanism_algorithms.jl:616-652 on the same kind of mechanism (10 revolute + 10 prismatic joints, where q̇ = v):
    with C = ½ ∂c/∂v (c = inverse_dynamics(v̇ = 0)) and Ṁ = Σ ∂M/∂q_k q̇_k, Ṁ - 2C is skew-symmetric."""
    rng = np.random.default_rng(36)
    model = rbd.flatten(rbd.rand_tree_mechanism(rng, ["Revolute"] * 10 + ["Prismatic"] * 10))
    q, v, _ = rand_inputs(rbd, model, 1, 37)
    nv = model.nv
    h = 1e-5

    def sym(M):
        return np.tril(M) + np.tril(M, -1).T

    Mdot = np.zeros((nv, nv))
    for k in range(nv):
        dq = np.zeros_like(q); dq[0, k] = h
        Mdot += (sym(oracle.mass_matrix(model, q + dq)[0]) - sym(oracle.mass_matrix(model, q - dq)[0])) / (2 * h) * v[0, k]
    C = np.zeros((nv, nv))
    zero = np.zeros_like(v)
    for k in range(nv):
        dv = np.zeros_like(v); dv[0, k] = h
        C[:, k] = (oracle.inverse_dynamics(model, q, v + dv, zero)[0] - oracle.inverse_dynamics(model, q, v - dv, zero)[0]) / (2 * h)
    C *= 0.5
    skew = Mdot - 2 * C
    scale = max(1.0, np.abs(Mdot).max(), np.abs(C).max())
    assert np.abs(skew + skew.T).max() <= 1e-6 * scale


def rand_floating_tree(rbd, seed):
    """rand_floating_tree_mechanism (src/mechanism_modification.jl:416-426) with the joint mix of the reference test: every body
    descends from one QuaternionFloating joint on the world."""
    rng = np.random.default_rng(seed)
    non_root = lambda mech, r: mech.bodies[1 + r.integers(len(mech.bodies) - 1)]
    return rbd.flatten(rbd.rand_tree_mechanism(rng, ["QuaternionFloating"] + ["Revolute"] * 10 + ["Planar"] * 10 + ["SinCosRevolute"] * 5, non_root))


@pytest.mark.parametrize("name", ["rand_floating_tree", "atlas_floating", "valkyrie_floating"])
def test_inverse_dynamics_external_wrenches_momentum_rate(rbd, oracle, models, name):
    """test/test_mechanism_algorithms.jl:707-727: for a mechanism on a floating joint, the floating joint's wrench (its rows of
    inverse_dynamics, moved to the root frame) + gravity + the external wrenches is the rate of change of the total momentum
    A(q) v.  Ties inverse_dynamics! with wrenches, momentum_matrix!, momentum_rate_bias, center_of_mass and the transforms together."""
    model = rand_floating_tree(rbd, 39) if name == "rand_floating_tree" else models[name]
    assert int(model.joint_type[0]) == 3 and list(model.parent).count(-1) == 1  # one body on the world, through the QuaternionFloating joint
    B = 3
    q, v, _, fe = rand_inputs(rbd, model, B, 39, fext=True)
    vd = np.random.default_rng(40).random((B, model.nv))
    tau = oracle.inverse_dynamics(model, q, v, vd, fe)
    H = oracle.transforms(model, q)[:, 0]  # the floating body's frame = frame_after of its joint
    R, p = H[:, :9].reshape(B, 3, 3), H[:, 9:]
    Rt, Rf = np.einsum("bij,bj->bi", R, tau[:, 0:3]), np.einsum("bij,bj->bi", R, tau[:, 3:6])
    total = np.concatenate([Rt + np.cross(p, Rf), Rf], axis=1)  # wrench transform (torque; force)
    _, _, com = oracle.momentum_matrix(model, q, v)
    mg = float(np.sum(model.inertia_mass)) * np.asarray(model.gravity, float)
    total += np.concatenate([np.cross(com, mg[None, :]), np.tile(mg, (B, 1))], axis=1)
    total += fe.reshape(B, model.n_bodies, 6).sum(axis=1)
    # the reference's own formula (:719): ḣ = Wrench(momentum_matrix, v̇) + momentum_rate_bias, atol 1e-10
    A, hsum, _ = oracle.momentum_matrix(model, q, v)
    hmom, hbias = oracle.momentum(model, q, v)
    assert np.abs(hmom - hsum).max() <= 1e-12 * max(1.0, np.abs(hsum).max())  # momentum(state) = Σ I_b T_b = A v (:527-545)
    hdot = np.einsum("bkn,bn->bk", A, vd) + hbias
    assert np.abs(total - hdot).max() <= 1e-10 * max(1.0, np.abs(hdot).max())
    # and the rate really is the time derivative of the momentum along (q̇, v̇) (central difference)
    _, qd = oracle.dynamics(model, q, v, want_qdot=True)
    h = 1e-6
    _, hp, _ = oracle.momentum_matrix(model, q + h * qd, v + h * vd)
    _, hm, _ = oracle.momentum_matrix(model, q - h * qd, v - h * vd)
    assert np.abs((hp - hm) / (2 * h) - hdot).max() <= 1e-6 * max(1.0, np.abs(hdot).max())


@pytest.mark.parametrize("name", ["atlas_floating", "randmech1", "randmech2", "inner_floating"])
def test_spatial_accelerations_are_twist_derivatives(rbd, oracle, models, name):
    """test/test_mechanism_algorithms.jl:459-490 (relative_acceleration vs the autodiff of relative_twist), with the root as base and central
    differences for the dual numbers: along q(t) = global_coordinates(q, t ϕ̇), v(t) = v + t v̇ the root-frame twist of every body changes at the
    rate spatial_accelerations! reports; inverse_dynamics!'s `accelerations` are those plus the root's −gravity (:387-417)."""
    import importlib
    sim = importlib.import_module("simulate_np")
    m = models[name]
    q, v, _ = rand_inputs(rbd, m, 1, 29)
    q, v = q[0], v[0]
    vd = np.random.default_rng(29).random(m.nv)
    _, _, A = oracle.body_kinematics(m, q[None], v[None], vd[None])
    h = 1e-6
    phid = sim.local_rate(m, q, q, v)  # the local-coordinate rates at ϕ = 0 (= v but for Planar joints, whose rates are world-aligned)
    qp, qm = sim.global_coordinates(m, q, h * phid), sim.global_coordinates(m, q, -h * phid)
    _, Tp, _ = oracle.body_kinematics(m, qp[None], (v + h * vd)[None], vd[None])
    _, Tm, _ = oracle.body_kinematics(m, qm[None], (v - h * vd)[None], vd[None])
    fd = (Tp - Tm) / (2 * h)
    assert np.abs(fd - A).max() <= 1e-7 * max(1.0, np.abs(A).max()), np.abs(fd - A).max()
    _, _, acc = oracle.inverse_dynamics_bodies(m, q[None], v[None], vd[None])
    g = np.asarray(m.gravity, float)
    assert np.abs(acc[0] - A[0] - np.r_[np.zeros(3), -g]).max() <= 1e-12 * max(1.0, np.abs(A).max())


@pytest.mark.parametrize("name", ["atlas_floating", "randmech1", "randmech3", "inner_floating"])
def test_momentum_rate_is_A_vdot_plus_bias(rbd, oracle, models, name):
    """test/test_mechanism_algorithms.jl:677-705: momentum two ways (A v = Σ I_b T_b), and its rate of change along a trajectory — central
    differences of momentum(q(t), v(t)) for the reference's dual numbers — equals momentum_matrix · v̇ + momentum_rate_bias."""
    import importlib
    sim = importlib.import_module("simulate_np")
    m = models[name]
    q, v, _ = rand_inputs(rbd, m, 1, 38)
    q, v = q[0], v[0]
    vd = np.random.default_rng(38).random(m.nv)
    A, hsum, _ = oracle.momentum_matrix(m, q[None], v[None])
    h0, bias = oracle.momentum(m, q[None], v[None])
    assert np.abs(A[0] @ v - h0[0]).max() <= 1e-12 * max(1.0, np.abs(h0).max()) and np.abs(hsum - h0).max() <= 1e-12 * max(1.0, np.abs(h0).max())
    h = 1e-6
    phid = sim.local_rate(m, q, q, v)
    hp, _ = oracle.momentum(m, sim.global_coordinates(m, q, h * phid)[None], (v + h * vd)[None])
    hm, _ = oracle.momentum(m, sim.global_coordinates(m, q, -h * phid)[None], (v - h * vd)[None])
    rate = A[0] @ vd + bias[0]
    assert np.abs((hp - hm)[0] / (2 * h) - rate).max() <= 1e-7 * max(1.0, np.abs(rate).max())


def test_remove_fixed_tree_joints_keeps_the_mass_matrix_and_the_contact_points(rbd, oracle):
    """test/test_mechanism_modification.jl:114-143: a random tree of 6-dof, revolute, spherical, planar, sin-cos AND fixed joints with 1-3 contact
    points per body; after remove_fixed_tree_joints! the tree joints are the non-fixed ones in their old order, no contact point is lost, and
    mass_matrix at the same q is unchanged to 1e-12 (pins the host-side merge of bodies the flattener relies on: src/mechanism_modification.jl:260-317)."""
    rng = np.random.default_rng(49)
    types = ["QuaternionFloating"] + ["Revolute"] * 10 + ["QuaternionSpherical", "Planar"] + ["Fixed"] * 10 + ["SinCosRevolute"] * 5
    rng.shuffle(types)
    mech = rbd.rand_tree_mechanism(rng, list(types))
    model = rbd.SoftContactModel(rbd.hunt_crossley_hertz(), rbd.ViscoelasticCoulombModel(0.8, 20e3, 100.0))
    npts = 0
    for body in mech.bodies[1:]:
        for _ in range(int(rng.integers(1, 4))):
            rbd.add_contact_point_(body, rbd.ContactPoint(rng.random(3), model, body.default_frame))
            npts += 1
    before = rbd.flatten(mech)
    q = rbd.rand_configuration(before, 3, rng)
    M0 = oracle.mass_matrix(before, q)
    nonfixed = [j.name for j in mech.tree_joints if j.joint_type.nv > 0]  # a Fixed joint is the one type without velocities
    rbd.remove_fixed_tree_joints_(mech)
    assert [j.name for j in mech.tree_joints] == nonfixed
    assert sum(len(b.contact_points) for b in mech.bodies) == npts
    mech.bodies[0].contact_points.clear()  # bodies that were fixed to the world left theirs on the root, where they act on nothing (the flat model takes none there)
    after = rbd.flatten(mech)
    assert after.nq == before.nq and after.nv == before.nv and after.n_bodies == len(nonfixed)
    M1 = oracle.mass_matrix(after, q)
    assert np.abs(np.tril(M1) - np.tril(M0)).max() <= 1e-12 * max(1.0, np.abs(M0).max())
    # ... and the points that are left sit where they sat in the world (their locations were re-expressed in the predecessor's frame)
    def world_points(flat):
        H = oracle.transforms(flat, q[:1])[0]
        pts = [H[c["body"]][:9].reshape(3, 3) @ c["location"] + H[c["body"]][9:] for c in flat.contact_points]
        return np.array(sorted(map(tuple, np.round(pts, 9))))
    kept = world_points(after)
    was = world_points(before)
    assert len(kept) <= len(was) and all(np.abs(was - p).sum(axis=1).min() < 1e-8 for p in kept)
