"""TEST INFRASTRUCTURE, NOT PRODUCT CODE — numpy restatement of the reference's `simulate` path:

  simulate(state0, final_time; Δt)             src/simulate.jl:36-55  (zero_torque! control, RK4 tableau, every step stored)
  MuntheKaasIntegrator.step                    src/ode_integrators.jl:233-299
  runge_kutta_4                                src/ode_integrators.jl:48-55
  local_coordinates! / global_coordinates!     src/joint_types/joint_types.jl:9-18 (default), sin_cos_revolute.jl:173-196,
                                               quaternion_spherical.jl:139-154, quaternion_floating.jl:205-249
  log_with_time_derivative / exp on SE(3)      src/spatial/spatialmotion.jl:226-332
  rotation_vector_rate                         src/spatial/util.jl:88-102

The dynamics inside each stage comes from the C oracle (oracle/rbd_oracle.c).  Rotations.jl conversions
(QuatRotation <-> RotationVec / AngleAxis / RotMatrix) are third-party and restated with quaternion algebra: within a
step the relative rotation is small, so every conversion is on its principal branch and the result is unique up to the
sign of the quaternion (tests compare rotation matrices).  One state at a time (small batches only)."""
from __future__ import annotations

import numpy as np

import oracle

FIXED, REVOLUTE, PRISMATIC, FLOATING, PLANAR, SPHERICAL, SINCOS = range(7)
NQ = {FIXED: 0, REVOLUTE: 1, PRISMATIC: 1, FLOATING: 7, PLANAR: 3, SPHERICAL: 4, SINCOS: 2}
NV = {FIXED: 0, REVOLUTE: 1, PRISMATIC: 1, FLOATING: 6, PLANAR: 3, SPHERICAL: 3, SINCOS: 1}
EPS = np.finfo(np.float64).eps


def qmul(a, b):
    w1, x1, y1, z1 = a
    w2, x2, y2, z2 = b
    return np.array([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                     w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2])


def qconj(a):
    return np.array([a[0], -a[1], -a[2], -a[3]])


def qrot(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def quat_from_rotvec(r):
    th = np.linalg.norm(r)
    s = 0.5 if th < EPS else np.sin(th / 2) / th
    return np.array([np.cos(th / 2), s * r[0], s * r[1], s * r[2]])


def rotvec_from_quat(q):
    s = np.linalg.norm(q[1:])
    th = 2 * np.arctan2(s, q[0])
    sc = 2.0 if s < EPS else th / s
    return sc * q[1:]


def se3_comm(xw, xv, yw, yv):
    return np.cross(xw, yw), np.cross(xw, yv) + np.cross(xv, yw)


def se3_log_with_rate(dq, dp, w, v):
    """_log + log_with_time_derivative (spatialmotion.jl:226-304) of the relative transform (dq, dp) with body twist (w, v)."""
    psi = rotvec_from_quat(dq)
    th = np.linalg.norm(psi)
    if th < EPS:
        return psi, dp.copy(), w.copy(), v.copy()
    th2 = th / 2
    alpha = th2 * np.cos(th2) / np.sin(th2)
    qv = dp - np.cross(psi, dp) / 2 + (1 - alpha) / th ** 2 * np.cross(psi, np.cross(psi, dp))
    beta = th2 ** 2 / np.sin(th2) ** 2
    A = (2 * (1 - alpha) + (alpha - beta) / 2) / th ** 2
    Bc = ((1 - alpha) + (alpha - beta) / 2) / th ** 4
    a1 = se3_comm(psi, qv, w, v)
    a2 = se3_comm(psi, qv, *a1)
    a3 = se3_comm(psi, qv, *a2)
    a4 = se3_comm(psi, qv, *a3)
    return psi, qv, w + a1[0] / 2 + A * a2[0] + Bc * a4[0], v + a1[1] / 2 + A * a2[1] + Bc * a4[1]


def se3_exp(prot, ptrans):
    """exp(::Twist) spatialmotion.jl:311-332 -> (relative quaternion, relative translation)."""
    th = np.linalg.norm(prot)
    if th < EPS:
        return np.array([1.0, 0, 0, 0]), ptrans.copy()
    om = prot / th
    dq = quat_from_rotvec(prot)
    vv = ptrans / th
    t = np.cross(om, vv)
    t = t - qrot(dq) @ t + om * (om @ vv) * th
    return dq, t


def rotation_vector_rate(phi, w):
    """spatial/util.jl:88-102 (Bortz equation)."""
    out = w + np.cross(phi, w) / 2
    th = np.linalg.norm(phi)
    if th > EPS:
        out = out + 1 / th ** 2 * (1 - (th * np.sin(th)) / (2 * (1 - np.cos(th)))) * np.cross(phi, np.cross(phi, w))
    return out


def local_rate(model, q0, q, v):
    """ϕ̇ of `local_coordinates!(ϕ, ϕ̇, state, q0)` for every tree joint (ϕ itself is not needed by the integrator)."""
    out = np.zeros(model.nv)
    for i in range(model.n_bodies):
        t, qo, vo = int(model.joint_type[i]), int(model.q_offset[i]), int(model.v_offset[i])
        qi0, qi, vi = q0[qo:qo + NQ[t]], q[qo:qo + NQ[t]], v[vo:vo + NV[t]]
        if t in (REVOLUTE, PRISMATIC, SINCOS):
            out[vo] = vi[0]
        elif t == PLANAR:  # default: ϕ̇ = velocity_to_configuration_derivative(q, v)
            s, c = np.sin(qi[2]), np.cos(qi[2])
            out[vo:vo + 3] = [c * vi[0] - s * vi[1], s * vi[0] + c * vi[1], vi[2]]
        elif t == SPHERICAL:
            phi = rotvec_from_quat(qmul(qconj(qi0), qi))
            out[vo:vo + 3] = rotation_vector_rate(phi, vi)
        elif t == FLOATING:
            dq = qmul(qconj(qi0[:4]), qi[:4])
            dp = qrot(qi0[:4]).T @ (qi[4:] - qi0[4:])
            _, _, wd, vd = se3_log_with_rate(dq, dp, vi[:3], vi[3:])
            out[vo:vo + 6] = np.r_[wd, vd]
    return out


def global_coordinates(model, q0, phi):
    q = np.zeros(model.nq)
    for i in range(model.n_bodies):
        t, qo, vo = int(model.joint_type[i]), int(model.q_offset[i]), int(model.v_offset[i])
        qi0, ph = q0[qo:qo + NQ[t]], phi[vo:vo + NV[t]]
        if t in (REVOLUTE, PRISMATIC, PLANAR):
            q[qo:qo + NQ[t]] = qi0 + ph
        elif t == SINCOS:
            s0, c0 = qi0
            sd, cd = np.sin(ph[0]), np.cos(ph[0])
            q[qo:qo + 2] = [s0 * cd + c0 * sd, c0 * cd - s0 * sd]
        elif t == SPHERICAL:
            q[qo:qo + 4] = qmul(qi0, quat_from_rotvec(ph))
        elif t == FLOATING:
            dq, dt = se3_exp(ph[:3], ph[3:])
            q[qo:qo + 4] = qmul(qi0[:4], dq)
            q[qo + 4:qo + 7] = qi0[4:] + qrot(qi0[:4]) @ dt
    return q


RK4_A = np.array([[0, 0, 0, 0], [0.5, 0, 0, 0], [0, 0.5, 0, 0], [0, 0, 1.0, 0]])
RK4_B = np.array([1 / 6, 1 / 3, 1 / 3, 1 / 6])


def vdot(model, q, v, tau, stabilize):
    if model.n_loops:
        return oracle.dynamics_loops(model, q[None], v[None], None if tau is None else tau[None], stabilize=stabilize)["vdot"][0]
    return oracle.dynamics(model, q[None], v[None], None if tau is None else tau[None])[0]


RK4_C = np.array([0.0, 0.5, 0.5, 1.0])


def step(model, q0, v0, dt, tau=None, stabilize=True, control=None, t0=0.0):
    """One MuntheKaasIntegrator step with the RK4 tableau (ode_integrators.jl:233-299).  control(t, q, v) -> τ is called with every stage's time
    and state, as the reference's closure calls control!(τ, t, state) (src/simulate.jl:42-48)."""
    phids, vds = [], []
    for i in range(4):
        phi = sum((dt * RK4_A[i, j] * phids[j] for j in range(i) if RK4_A[i, j] != 0), np.zeros(model.nv))
        v = v0 + sum((dt * RK4_A[i, j] * vds[j] for j in range(i) if RK4_A[i, j] != 0), np.zeros(model.nv))
        q = global_coordinates(model, q0, phi)
        if control is not None:
            tau = control(t0 + RK4_C[i] * dt, q, v)
        vds.append(vdot(model, q, v, tau, stabilize))
        phids.append(local_rate(model, q0, q, v))
    phi = sum(dt * RK4_B[i] * phids[i] for i in range(4))
    v = v0 + sum(dt * RK4_B[i] * vds[i] for i in range(4))
    return global_coordinates(model, q0, phi), v


def simulate(model, q, v, final_time, dt, tau=None, stabilize=True, control=None):
    """`simulate`: steps while t < final_time (ode_integrators.jl:307-316). q, v: (B, n). Returns ts, q_end, v_end.
    control(b, t, q_b, v_b) -> τ_b: a controller evaluated at every stage of every state."""
    q, v = np.array(q, float), np.array(v, float)
    t, ts = 0.0, [0.0]
    while t < final_time:
        for b in range(q.shape[0]):
            cb = None if control is None else (lambda tt, qq, vv, b=b: control(b, tt, qq, vv))
            q[b], v[b] = step(model, q[b], v[b], dt, None if tau is None else tau[b], stabilize, cb, t)
        t += dt
        ts.append(t)
    return np.array(ts), q, v


def vdot_contact(model, q, v, s, tau):
    vd, _, sd, _, _ = oracle.dynamics_contact(model, q[None], v[None], s[None], None if tau is None else tau[None])
    return vd[0], sd[0]


def step_contact(model, q0, v0, s0, dt, tau=None):
    """MuntheKaasIntegrator.step with additional states (ode_integrators.jl:233-299): s goes through the tableau like v."""
    phids, vds, sds = [], [], []
    for i in range(4):
        phi = sum((dt * RK4_A[i, j] * phids[j] for j in range(i) if RK4_A[i, j] != 0), np.zeros(model.nv))
        v = v0 + sum((dt * RK4_A[i, j] * vds[j] for j in range(i) if RK4_A[i, j] != 0), np.zeros(model.nv))
        s = s0 + sum((dt * RK4_A[i, j] * sds[j] for j in range(i) if RK4_A[i, j] != 0), np.zeros(model.ns))
        q = global_coordinates(model, q0, phi)
        vd, sd = vdot_contact(model, q, v, s, tau)
        vds.append(vd); sds.append(sd)
        phids.append(local_rate(model, q0, q, v))
    phi = sum(dt * RK4_B[i] * phids[i] for i in range(4))
    v = v0 + sum(dt * RK4_B[i] * vds[i] for i in range(4))
    s = s0 + sum(dt * RK4_B[i] * sds[i] for i in range(4))
    return global_coordinates(model, q0, phi), v, s


def simulate_contact(model, q, v, s, final_time, dt, tau=None, record=False):
    """`simulate` of a mechanism with contact points. q, v, s: (B, n). Returns ts, q_end, v_end, s_end (and the trajectory when record)."""
    q, v, s = np.array(q, float), np.array(v, float), np.array(s, float)
    t, ts, traj = 0.0, [0.0], [(q.copy(), v.copy(), s.copy())]
    while t < final_time:
        for b in range(q.shape[0]):
            q[b], v[b], s[b] = step_contact(model, q[b], v[b], s[b], dt, None if tau is None else tau[b])
        t += dt
        ts.append(t)
        if record:
            traj.append((q.copy(), v.copy(), s.copy()))
    return (np.array(ts), q, v, s, traj) if record else (np.array(ts), q, v, s)


# ---- the same step, vectorised over the batch (whole-batch parity of `simulate` at 4096 … 65 536 states: bench.py, GPU tests) ---------------------------
# Same formulas as above, arrays shaped (B, …); mechanisms of Fixed / Revolute / Prismatic / QuaternionFloating tree joints without loop joints (Atlas,
# Valkyrie); anything else takes the one-state-at-a-time functions.  tests/test_oracle_simulate.py checks it against `step` state by state.
def _bq_mul(a, b):
    w1, x1, y1, z1 = a.T
    w2, x2, y2, z2 = b.T
    return np.stack([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                     w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2], axis=1)


def _bq_conj(a):
    return a * np.array([1.0, -1.0, -1.0, -1.0])


def _bq_rot(q):
    w, x, y, z = q.T
    return np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], axis=1),
                     np.stack([2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)], axis=1),
                     np.stack([2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], axis=1)], axis=1)


def _b_log_with_rate(dq, dp, w, v):
    s = np.linalg.norm(dq[:, 1:], axis=1)
    th = 2 * np.arctan2(s, dq[:, 0])
    small = th < EPS
    sc = np.where(s < EPS, 2.0, th / np.where(s < EPS, 1.0, s))
    psi = sc[:, None] * dq[:, 1:]
    ths = np.where(small, 1.0, th)  # (small angles: the branch that returns the inputs, below)
    th2 = ths / 2
    alpha = th2 * np.cos(th2) / np.sin(th2)
    qv = dp - np.cross(psi, dp) / 2 + ((1 - alpha) / ths ** 2)[:, None] * np.cross(psi, np.cross(psi, dp))
    beta = th2 ** 2 / np.sin(th2) ** 2
    A = (2 * (1 - alpha) + (alpha - beta) / 2) / ths ** 2
    Bc = ((1 - alpha) + (alpha - beta) / 2) / ths ** 4
    a1 = se3_comm(psi, qv, w, v)
    a2 = se3_comm(psi, qv, *a1)
    a3 = se3_comm(psi, qv, *a2)
    a4 = se3_comm(psi, qv, *a3)
    wd = w + a1[0] / 2 + A[:, None] * a2[0] + Bc[:, None] * a4[0]
    vd = v + a1[1] / 2 + A[:, None] * a2[1] + Bc[:, None] * a4[1]
    return np.where(small[:, None], w, wd), np.where(small[:, None], v, vd)


def _b_exp(prot, ptrans):
    th = np.linalg.norm(prot, axis=1)
    small = th < EPS
    ths = np.where(small, 1.0, th)
    om = prot / ths[:, None]
    dq = np.concatenate([np.cos(ths / 2)[:, None], (np.sin(ths / 2) / ths)[:, None] * prot], axis=1)
    vv = ptrans / ths[:, None]
    t = np.cross(om, vv)
    t = t - np.einsum("bij,bj->bi", _bq_rot(dq), t) + om * (np.einsum("bi,bi->b", om, vv) * ths)[:, None]
    dq = np.where(small[:, None], np.array([1.0, 0, 0, 0]), dq)
    return dq, np.where(small[:, None], ptrans, t)


def batchable(model):
    return model.n_loops == 0 and all(int(t) in (FIXED, REVOLUTE, PRISMATIC, FLOATING) for t in model.joint_type)


def local_rate_batch(model, q0, q, v):
    out = np.array(v, float)  # 1-dof joints: ϕ̇ = v
    for i in range(model.n_bodies):
        if int(model.joint_type[i]) == FLOATING:
            qo, vo = int(model.q_offset[i]), int(model.v_offset[i])
            dq = _bq_mul(_bq_conj(q0[:, qo:qo + 4]), q[:, qo:qo + 4])
            dp = np.einsum("bji,bj->bi", _bq_rot(q0[:, qo:qo + 4]), q[:, qo + 4:qo + 7] - q0[:, qo + 4:qo + 7])
            wd, vd = _b_log_with_rate(dq, dp, v[:, vo:vo + 3], v[:, vo + 3:vo + 6])
            out[:, vo:vo + 3], out[:, vo + 3:vo + 6] = wd, vd
    return out


def global_coordinates_batch(model, q0, phi):
    q = np.zeros_like(q0)
    for i in range(model.n_bodies):
        t, qo, vo = int(model.joint_type[i]), int(model.q_offset[i]), int(model.v_offset[i])
        if t in (REVOLUTE, PRISMATIC):
            q[:, qo] = q0[:, qo] + phi[:, vo]
        elif t == FLOATING:
            dq, dt = _b_exp(phi[:, vo:vo + 3], phi[:, vo + 3:vo + 6])
            q[:, qo:qo + 4] = _bq_mul(q0[:, qo:qo + 4], dq)
            q[:, qo + 4:qo + 7] = q0[:, qo + 4:qo + 7] + np.einsum("bij,bj->bi", _bq_rot(q0[:, qo:qo + 4]), dt)
    return q


def step_batch(model, q0, v0, dt, tau=None, nthreads=1):
    """`step` for every state of a batch at once (constant τ): the four dynamics evaluations are one batched oracle call each."""
    assert batchable(model)
    phids, vds = [], []
    for i in range(4):
        phi = sum((dt * RK4_A[i, j] * phids[j] for j in range(i) if RK4_A[i, j] != 0), np.zeros_like(v0))
        v = v0 + sum((dt * RK4_A[i, j] * vds[j] for j in range(i) if RK4_A[i, j] != 0), np.zeros_like(v0))
        q = global_coordinates_batch(model, q0, phi)
        vds.append(oracle.dynamics(model, q, v, tau, nthreads=nthreads))
        phids.append(local_rate_batch(model, q0, q, v))
    phi = sum(dt * RK4_B[i] * phids[i] for i in range(4))
    v = v0 + sum(dt * RK4_B[i] * vds[i] for i in range(4))
    return global_coordinates_batch(model, q0, phi), v


def simulate_batch(model, q, v, nsteps, dt, tau=None, nthreads=1):
    q, v = np.array(q, float), np.array(v, float)
    for _ in range(nsteps):
        q, v = step_batch(model, q, v, dt, tau, nthreads)
    return q, v
