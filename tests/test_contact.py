"""Soft contact: contact_dynamics! (src/mechanism_algorithms.jl:680-723) with the reference's default point model
(src/contact.jl: Hunt–Crossley normal force, viscoelastic Coulomb friction, half-space environment) and `simulate` with the additional
state integrated beside (q, v).

  CPU: the oracle's restatement (oracle/rbd_oracle_impl.h::rbdo_contact_dynamics + oracle/simulate_np.py::simulate_contact) against the
       reference's own known-answer tests — the elastic ball drop (energy balance incl. the elastic potential, bounces:
       test/test_simulate.jl:36-84) and the inclined plane (stick above the critical friction coefficient, slip below: :86-125).
  GPU: rbd_contact_dynamics / rbd_dynamics_contact / rbd_simulate_contact through the C ABI against that oracle."""
import numpy as np
import pytest


def ball(rbd, rng, alpha=0.0):
    """test/test_simulate.jl:40-57: one rigid body with a random inertia on a floating joint, a contact point at its centre of mass,
    the floor z = 0."""
    world = rbd.RigidBody("world")
    mech = rbd.Mechanism(world)
    frame = rbd.CartesianFrame3D("body")
    A = rng.standard_normal((3, 3))
    com = 0.1 * rng.standard_normal(3)
    body = rbd.RigidBody("body", rbd.SpatialInertia(frame, mass=1.0 + rng.random(), com=com, moment_about_com=A @ A.T + 0.5 * np.eye(3)))
    joint = rbd.Joint("floating", rbd.QuaternionFloating())
    rbd.attach_(mech, world, body, joint)
    com_now = body.inertia.cross_part / body.inertia.mass  # centre of mass in the body's (canonicalised) default frame
    model = rbd.SoftContactModel(rbd.hunt_crossley_hertz(alpha=alpha), rbd.ViscoelasticCoulombModel(0.5, 1e3, 1e3))
    rbd.add_contact_point_(body, rbd.ContactPoint(com_now, model))
    rbd.add_environment_primitive_(mech, rbd.HalfSpace3D([0, 0, 0], [0, 0, 1.0]))
    return mech, com_now, model


def com_height_energy(oracle, flat, com, model, q, v):
    H = oracle.transforms(flat, q)[:, 0]
    R, p = H[:, :9].reshape(-1, 3, 3), H[:, 9:]
    z = (R @ com + p)[:, 2]
    ke, pe = oracle.energy(flat, q, v)
    pen = np.maximum(-z, 0.0)
    n = model.normal.n
    return z, model.normal.k * pen ** (n + 1) / (n + 1) + ke + pe


def test_oracle_elastic_ball_drop(rbd, oracle):
    """Energy (kinetic + gravitational + elastic potential k z^(n+1)/(n+1)) stays within 1e-2 of its initial value over 0.5 s of bouncing on
    a conservative (α = 0) floor, and the vertical velocity of the contact point changes sign more than 3 times (test_simulate.jl:59-84)."""
    import simulate_np
    mech, com, model = ball(rbd, np.random.default_rng(61))
    flat = rbd.flatten(mech)
    assert flat.ns == 3
    z0 = 0.05
    q = np.array([[1.0, 0, 0, 0, 1.0, 2.0, z0 - com[2]]])
    v = np.zeros((1, 6))
    _, e0 = com_height_energy(oracle, flat, com, model, q, v)
    ts, qe, ve, se, traj = simulate_np.simulate_contact(flat, q, v, np.zeros((1, 3)), 0.5, 1e-3, record=True)
    sign, switches = 0.0, 0
    for (qk, vk, sk) in traj:
        z, e = com_height_energy(oracle, flat, com, model, qk, vk)
        assert abs(e[0] - e0[0]) <= 1e-2
        _, T, _ = oracle.body_kinematics(flat, qk, vk, np.zeros_like(vk))
        H = oracle.transforms(flat, qk)[0, 0]
        pt = H[:9].reshape(3, 3) @ com + H[9:]
        vz = (np.cross(T[0, 0, :3], pt) + T[0, 0, 3:])[2]
        new = np.sign(vz)
        switches += new != sign
        sign = new
    assert switches > 3


def test_oracle_inclined_plane(rbd, oracle):
    """A point mass on a plane inclined by θ = 0.5: with μ just above tan θ it sticks (moves < 1e-4 in 0.5 s after settling), just below it
    slides (> 5e-2); a second, far-away half-space must not matter (test_simulate.jl:86-125, issue #211)."""
    import simulate_np
    theta = 0.5
    mucrit = np.tan(theta)
    moved = {}
    for mu in (mucrit + 1e-2, mucrit - 1e-2):
        world = rbd.RigidBody("world")
        mech = rbd.Mechanism(world)
        body = rbd.RigidBody("body", rbd.SpatialInertia(rbd.CartesianFrame3D("inertia"), moment=np.eye(3), cross_part=np.zeros(3), mass=2.0))
        rbd.attach_(mech, world, body, rbd.Joint("floating", rbd.QuaternionFloating()))
        rbd.add_environment_primitive_(mech, rbd.HalfSpace3D([0, 0, 0], [np.sin(theta), 0, np.cos(theta)]))
        rbd.add_environment_primitive_(mech, rbd.HalfSpace3D([0, 0, -100.0], [0, 0, 1.0]))
        model = rbd.SoftContactModel(rbd.hunt_crossley_hertz(k=50e3, alpha=1.0), rbd.ViscoelasticCoulombModel(mu, 50e3, 1e4))
        rbd.add_contact_point_(body, rbd.ContactPoint(np.zeros(3), model))
        flat = rbd.flatten(mech)
        assert flat.ns == 6
        q = np.array([[1.0, 0, 0, 0, 0, 0, 0]])
        _, q1, v1, s1 = simulate_np.simulate_contact(flat, q, np.zeros((1, 6)), np.zeros((1, 6)), 1.0, 1e-3)
        _, q2, v2, s2 = simulate_np.simulate_contact(flat, q1, v1, s1, 0.5, 1e-3)
        moved[mu] = np.abs(q2[0, 4:] - q1[0, 4:]).max()
        assert np.all(s2[0, 3:] == 0)  # never in contact with the far plane: its state stays reset
    assert moved[mucrit + 1e-2] <= 1e-4
    assert moved[mucrit - 1e-2] > 5e-2


def walker(rbd, rng):
    """A small floating tree with contact points on several bodies (two on one of them) and two half-spaces."""
    mech = rbd.rand_tree_mechanism(rng, ["QuaternionFloating", "Revolute", "Revolute", "Prismatic", "Revolute"])
    bodies = mech.bodies[1:]
    for b, npts in zip((bodies[0], bodies[2], bodies[4]), (1, 2, 1)):
        for _ in range(npts):
            model = rbd.SoftContactModel(rbd.hunt_crossley_hertz(k=2e3 * (1 + rng.random()), alpha=0.3 * rng.random()),
                                         rbd.ViscoelasticCoulombModel(0.3 + rng.random(), 1e3 * (1 + rng.random()), 1e2 * (1 + rng.random())))
            rbd.add_contact_point_(b, rbd.ContactPoint(0.3 * rng.standard_normal(3), model))
    rbd.add_environment_primitive_(mech, rbd.HalfSpace3D([0, 0, 0.2], [0.1, -0.2, 1.0]))
    rbd.add_environment_primitive_(mech, rbd.HalfSpace3D([0.3, 0, 0], [1.0, 0.3, 0.1]))
    return mech


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["aos", "soa"])
def test_contact_dynamics_matches_oracle(rbd, oracle, layout):
    """contact_dynamics! and dynamics! with contact points on random states (some points in contact, some not) against the oracle: contact
    wrenches, total wrenches, ṡ, the reset of s, v̇ — fp64 at 1e-10."""
    import torch
    rng = np.random.default_rng(5)
    flat = rbd.flatten(walker(rbd, rng))
    assert flat.ns == 3 * 4 * 2
    B = 130
    q, v = rbd.rand_configuration(flat, B, rng), rbd.rand_velocity(flat, B, rng)
    q[:, 4:7] *= 0.5
    s = 1e-3 * rng.standard_normal((B, flat.ns))
    tau, fe = rng.random((B, flat.nv)), rng.random((B, 6 * flat.n_bodies))
    vd_ref, s_ref, sd_ref, cw_ref, tw_ref = oracle.dynamics_contact(flat, q, v, s, tau, fe)
    inside = (sd_ref.reshape(B, -1, 3) != 0).any(axis=2)
    assert inside.any() and not inside.all()  # the random states exercise both branches
    state = rbd.MechanismState(flat, B, layout=layout)
    rbd.set_configuration_(state, q); rbd.set_velocity_(state, v)
    dev = lambda a: torch.as_tensor(np.ascontiguousarray(a if layout == "aos" else a.T)).cuda()
    host = lambda t: (t if layout == "aos" else t.t()).cpu().numpy()
    state.s.copy_(dev(s))
    result = rbd.DynamicsResult(flat, B, layout=layout)
    rbd.dynamics_(result, state, dev(tau), dev(fe))
    assert rbd.sync(state) == 0
    rel = lambda a, b: np.abs(a - b).max() / max(1.0, np.abs(b).max())
    assert rel(host(result.contactwrenches), cw_ref) <= 1e-10
    assert rel(host(result.totalwrenches), tw_ref) <= 1e-10
    assert rel(host(result.sd), sd_ref) <= 1e-10
    assert np.array_equal(host(state.s) == 0, s_ref == 0) and rel(host(state.s), s_ref) <= 1e-14
    assert rel(host(result.vd), vd_ref) <= 1e-10


@pytest.mark.gpu
def test_contact_dynamics_on_a_tree_of_more_than_64_bodies(rbd, oracle):
    """Round 4: contact points on a tree the wavefront-shaped kernels do not take (70 bodies): contact_dynamics! and dynamics! through the any-size kernels
    (per-body kinematics from big_rnea_kernel's scratch, then the same contact kernel, then the reference's CRBA + Cholesky route) against the oracle."""
    import torch
    rng = np.random.default_rng(15)
    mech = rbd.rand_tree_mechanism(rng, ["QuaternionFloating"] + ["Revolute"] * 69)
    for b in mech.bodies[1:]:
        if rng.random() < 0.15:
            model = rbd.SoftContactModel(rbd.hunt_crossley_hertz(k=2e3 * (1 + rng.random()), alpha=0.3 * rng.random()),
                                         rbd.ViscoelasticCoulombModel(0.3 + rng.random(), 1e3 * (1 + rng.random()), 1e2 * (1 + rng.random())))
            rbd.add_contact_point_(b, rbd.ContactPoint(0.3 * rng.standard_normal(3), model))
    rbd.add_environment_primitive_(mech, rbd.HalfSpace3D([0, 0, 0.2], [0.1, -0.2, 1.0]))
    flat = rbd.flatten(mech)
    assert flat.n_bodies == 70 and flat.ns > 0
    B = 21
    q, v = rbd.rand_configuration(flat, B, rng), rbd.rand_velocity(flat, B, rng)
    s = 1e-3 * rng.standard_normal((B, flat.ns))
    tau, fe = rng.random((B, flat.nv)), rng.random((B, 6 * flat.n_bodies))
    vd_ref, s_ref, sd_ref, cw_ref, tw_ref = oracle.dynamics_contact(flat, q, v, s, tau, fe)
    inside = (sd_ref.reshape(B, -1, 3) != 0).any(axis=2)
    assert inside.any() and not inside.all()
    state = rbd.MechanismState(flat, B)
    rbd.set_configuration_(state, q); rbd.set_velocity_(state, v)
    dev = lambda a: torch.as_tensor(np.ascontiguousarray(a)).cuda()
    state.s.copy_(dev(s))
    result = rbd.DynamicsResult(flat, B)
    rbd.dynamics_(result, state, dev(tau), dev(fe))
    assert rbd.sync(state) == 0
    rel = lambda a, b: np.abs(a - b).max() / max(1.0, np.abs(b).max())
    assert rel(result.contactwrenches.cpu().numpy(), cw_ref) <= 1e-10
    assert rel(result.totalwrenches.cpu().numpy(), tw_ref) <= 1e-10
    assert rel(result.sd.cpu().numpy(), sd_ref) <= 1e-10
    assert rel(state.s.cpu().numpy(), s_ref) <= 1e-14
    cond = np.linalg.cond(np.tril(oracle.mass_matrix(flat, q)) + np.transpose(np.tril(oracle.mass_matrix(flat, q), -1), (0, 2, 1))).max()
    assert rel(result.vd.cpu().numpy(), vd_ref) <= 1e-14 * cond


@pytest.mark.gpu
def test_simulate_contact_ball_drop_batch(rbd, oracle):
    """`simulate` with contact on the GPU: a batch of balls dropped from different heights.  Against the numpy restatement after 120 steps
    (through first impact for the lowest drops) at 1e-9, and the energy balance of the reference's test along the whole 0.5 s."""
    import torch
    import simulate_np
    mech, com, model = ball(rbd, np.random.default_rng(61))
    flat = rbd.flatten(mech)
    B = 9
    q = np.tile(np.array([1.0, 0, 0, 0, 1.0, 2.0, 0.0]), (B, 1))
    q[:, 6] = np.linspace(0.002, 0.06, B) - com[2]
    v = np.zeros((B, 6))
    _, e0 = com_height_energy(oracle, flat, com, model, q, v)
    state = rbd.MechanismState(flat, B)
    rbd.set_configuration_(state, q); rbd.set_velocity_(state, v)
    rbd.simulate_(state, 0.1195, dt=1e-3)
    _, q_ref, v_ref, s_ref = simulate_np.simulate_contact(flat, q, v, np.zeros((B, 3)), 0.1195, 1e-3)
    qg, vg, sg = state.q.cpu().numpy(), state.v.cpu().numpy(), state.s.cpu().numpy()
    assert np.abs(qg[:, 4:] - q_ref[:, 4:]).max() <= 1e-9 and np.abs(np.abs(qg[:, :4]) - np.abs(q_ref[:, :4])).max() <= 1e-9
    assert np.abs(vg - v_ref).max() <= 1e-8 * max(1.0, np.abs(v_ref).max())
    assert np.abs(sg - s_ref).max() <= 1e-9
    ts, qs, vs, ss = rbd.simulate_(state, 0.38, dt=1e-3, store=True)
    for qk, vk in zip(qs[::10], vs[::10]):
        _, e = com_height_energy(oracle, flat, com, model, qk.cpu().numpy(), vk.cpu().numpy())
        assert np.abs(e - e0).max() <= 1e-2
    assert rbd.sync(state) == 0


@pytest.mark.gpu
def test_contact_entry_points_reject_what_they_cannot_do(rbd, models):
    import ctypes, torch
    from rigidbodydynamics_jl_amd import _capi
    state = rbd.MechanismState(models["double_pendulum"], 4)
    z = torch.zeros(4, 8, dtype=torch.float64, device="cuda")
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    st = _capi.lib().rbd_contact_dynamics(state.ws.handle, 4, p(state.q), p(state.v), p(z), None, p(z), ctypes.byref(state._opts()))
    assert st == 1  # RBD_ERR_INVALID_ARGUMENT: the mechanism has no contact points


@pytest.mark.gpu
def test_contact_ode_form_and_result_fields(rbd, oracle):
    """Round-2 advisor findings: dynamics!(ẋ, result, state, x) of a mechanism with contact points takes x = [q; v; s] and returns ẋ = [q̇; v̇; ṡ]
    (src/mechanism_state.jl:419-426, src/dynamics_result.jl:89-95); a DynamicsResult with per-body fields gets the bias accelerations / joint
    wrenches computed WITH the total (external + contact) wrenches, and the "crba" route its mass matrix; the plain C entry points refuse the
    model instead of leaving the contact wrenches out."""
    import ctypes
    import torch
    rng = np.random.default_rng(15)
    flat = rbd.flatten(walker(rbd, rng))
    B = 37
    q, v = rbd.rand_configuration(flat, B, rng), rbd.rand_velocity(flat, B, rng)
    q[:, 4:7] *= 0.5
    s = 1e-3 * rng.standard_normal((B, flat.ns))
    tau, fe = rng.random((B, flat.nv)), rng.random((B, 6 * flat.n_bodies))
    vd_ref, s_ref, sd_ref, cw_ref, tw_ref = oracle.dynamics_contact(flat, q, v, s, tau, fe)
    state = rbd.MechanismState(flat, B)
    result = rbd.DynamicsResult(flat, B, bodies=True)
    x = torch.as_tensor(np.concatenate([q, v, s], axis=1)).cuda()
    xd = torch.zeros_like(x)
    rbd.dynamics_ode_(xd, result, state, x, torch.as_tensor(tau).cuda(), torch.as_tensor(fe).cuda())
    got = xd.cpu().numpy()
    rel = lambda a, b: np.abs(a - b).max() / max(1.0, np.abs(b).max())
    assert rel(got[:, flat.nq:flat.nq + flat.nv], vd_ref) <= 1e-10 and rel(got[:, flat.nq + flat.nv:], sd_ref) <= 1e-10
    with pytest.raises(rbd.DimensionMismatch):
        rbd.dynamics_ode_(xd[:, :flat.nq + flat.nv], result, state, x[:, :flat.nq + flat.nv])
    c_ref, jw_ref, acc_ref = oracle.inverse_dynamics_bodies(flat, q, v, None, tw_ref)
    assert rel(result.jointwrenches.cpu().numpy().reshape(B, -1, 6), jw_ref) <= 1e-10
    assert rel(result.accelerations.cpu().numpy().reshape(B, -1, 6), acc_ref) <= 1e-10
    assert rel(result.dynamicsbias.cpu().numpy(), c_ref) <= 1e-10
    state.s.copy_(torch.as_tensor(s).cuda())
    rbd.dynamics_(result, state, torch.as_tensor(tau).cuda(), torch.as_tensor(fe).cuda(), algorithm="crba")
    M = result.massmatrix.cpu().numpy().reshape(B, flat.nv, flat.nv).transpose(0, 2, 1)
    assert rel(np.tril(M), np.tril(oracle.mass_matrix(flat, q))) <= 1e-10
    # the C ABI without the additional state: refused (RBD_ERR_UNSUPPORTED = 3), not mis-evaluated
    from rigidbodydynamics_jl_amd import _capi
    L, vp = _capi.lib(), ctypes.c_void_p
    opts = state._opts(0)
    st = L.rbd_dynamics(state.ws.handle, B, vp(state.q.data_ptr()), vp(state.v.data_ptr()), None, None, vp(result.vd.data_ptr()), None, None, ctypes.byref(opts))
    assert st == 3
    st = L.rbd_simulate(state.ws.handle, B, vp(state.q.data_ptr()), vp(state.v.data_ptr()), None, None, ctypes.c_double(1e-3), 2, ctypes.byref(opts))
    assert st == 3
