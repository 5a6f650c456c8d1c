"""featherstone_np.py — an INDEPENDENT second statement of the hot path, for pinning the oracle.  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

`oracle/rbd_oracle_impl.h` restates the reference's own formulation line by line: every quantity in the ROOT frame, Transform3D
products, `newton_euler` on root-frame inertias (src/mechanism_state.jl:686-868, src/mechanism_algorithms.jl).  This module computes the
same τ = M(q) v̇ + c(q, v, w_ext), M(q) and v̇ = M⁻¹(τ − c) with the textbook formulation instead — R. Featherstone, *Rigid Body Dynamics
Algorithms* (2008): Table 5.1 (RNEA), Table 6.2 (CRBA), Table 7.1 (ABA) — in BODY coordinates with 6×6 Plücker transforms, written from
the book and sharing no helper, no intermediate and no frame convention with the C restatement.  Where the two agree to ~1e-12 on Atlas-size
mechanisms (tests/test_golden_vectors.py) a mistake would have to be made identically in two unrelated derivations.

Only the model *data* is shared (the flat model read from the reference's URDF fixtures) and the conventions that define the inputs and
outputs: q, v orderings of the reference's joint types (src/joint_types/*.jl), motion vectors (angular; linear), gravity as a field.
"""
import numpy as np

FIXED, REVOLUTE, PRISMATIC, QUAT_FLOATING, PLANAR, QUAT_SPHERICAL, SINCOS_REVOLUTE = range(7)
NQ = {FIXED: 0, REVOLUTE: 1, PRISMATIC: 1, QUAT_FLOATING: 7, PLANAR: 3, QUAT_SPHERICAL: 4, SINCOS_REVOLUTE: 2}
NV = {FIXED: 0, REVOLUTE: 1, PRISMATIC: 1, QUAT_FLOATING: 6, PLANAR: 3, QUAT_SPHERICAL: 3, SINCOS_REVOLUTE: 1}


def skew(a):
    return np.array([[0.0, -a[2], a[1]], [a[2], 0.0, -a[0]], [-a[1], a[0], 0.0]])


def plucker(E, r):
    """Motion transform A -> B for a frame B whose axes are E (B-coordinates = E · A-coordinates) and whose origin sits at r (A coordinates):
    X = [[E, 0], [−E r×, E]]   (RBDA eq. 2.24)."""
    X = np.zeros((6, 6))
    X[:3, :3] = E
    X[3:, 3:] = E
    X[3:, :3] = -E @ skew(r)
    return X


def crm(v):
    """v× for motion vectors (RBDA eq. 2.31)."""
    X = np.zeros((6, 6))
    X[:3, :3] = skew(v[:3])
    X[3:, 3:] = skew(v[:3])
    X[3:, :3] = skew(v[3:])
    return X


def crf(v):
    """v×* for force vectors."""
    return -crm(v).T


def rodrigues(a, s, c):
    """Rotation by the angle with sine s, cosine c about the unit axis a."""
    K = skew(a)
    return np.eye(3) + s * K + (1.0 - c) * (K @ K)


def quat_matrix(w, x, y, z):
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def joint_model(model, i, q):
    """(R_J, p_J, S): the joint places frame_after in frame_before by x_before = R_J x_after + p_J; S (6 × nv_i) is the joint's motion
    subspace in frame_after = body coordinates."""
    t = int(model.joint_type[i])
    qo = int(model.q_offset[i])
    a = np.asarray(model.joint_axis[i], float)
    R, p, S = np.eye(3), np.zeros(3), np.zeros((6, NV[t]))
    if t == REVOLUTE:
        R = rodrigues(a, np.sin(q[qo]), np.cos(q[qo]))
        S[:3, 0] = a
    elif t == SINCOS_REVOLUTE:
        R = rodrigues(a, q[qo], q[qo + 1])
        S[:3, 0] = a
    elif t == PRISMATIC:
        p = a * q[qo]
        S[3:, 0] = a
    elif t == QUAT_FLOATING:
        R = quat_matrix(*q[qo:qo + 4])
        p = np.array(q[qo + 4:qo + 7], float)
        S = np.eye(6)
    elif t == QUAT_SPHERICAL:
        R = quat_matrix(*q[qo:qo + 4])
        S[:3, :3] = np.eye(3)
    elif t == PLANAR:
        ax, ay = a, np.asarray(model.joint_axis2[i], float)
        az = np.cross(ax, ay)
        R = rodrigues(az, np.sin(q[qo + 2]), np.cos(q[qo + 2]))
        p = ax * q[qo] + ay * q[qo + 1]
        S[3:, 0], S[3:, 1], S[:3, 2] = ax, ay, az
    return R, p, S


def body_inertia(model, i):
    """6×6 spatial inertia about the body-frame origin (RBDA eq. 2.63): [[J, c×], [c×', m 1]], c = m · com."""
    I = np.zeros((6, 6))
    C = skew(np.asarray(model.inertia_cross[i], float))
    I[:3, :3] = np.asarray(model.inertia_moment[i], float).reshape(3, 3)
    I[:3, 3:] = C
    I[3:, :3] = C.T
    I[3:, 3:] = float(model.inertia_mass[i]) * np.eye(3)
    return I


class Kin:
    """Per-state kinematics in body coordinates: Xup[i] (parent body -> body i), X0[i] (root -> body i), S[i]."""

    def __init__(self, model, q):
        n = model.n_bodies
        self.Xup, self.X0, self.S = [None] * n, [None] * n, [None] * n
        for i in range(n):
            Rj, pj, S = joint_model(model, i, q)
            Rt, pt = np.asarray(model.pred_rot[i], float).reshape(3, 3), np.asarray(model.pred_trans[i], float)
            # predecessor body -> frame_before (joint_to_predecessor inverted), then frame_before -> frame_after (the joint)
            self.Xup[i] = plucker(Rj.T, pj) @ plucker(Rt.T, pt)
            p = int(model.parent[i])
            self.X0[i] = self.Xup[i] if p < 0 else self.Xup[i] @ self.X0[p]
            self.S[i] = S


def rnea(model, q, v, vd=None, fext=None):
    """RBDA Table 5.1.  Gravity enters as the base acceleration −a_g; fext[6 i : 6 i + 6] is the wrench on body i given in ROOT coordinates
    (the reference's externalwrenches), moved to body coordinates by X0⁻ᵀ."""
    n = model.n_bodies
    K = Kin(model, q)
    a0 = np.concatenate([np.zeros(3), -np.asarray(model.gravity, float)])
    vel, acc, f = [None] * n, [None] * n, [None] * n
    for i in range(n):
        p, vo, nv = int(model.parent[i]), int(model.v_offset[i]), NV[int(model.joint_type[i])]
        vJ = K.S[i] @ v[vo:vo + nv]
        aJ = K.S[i] @ (vd[vo:vo + nv] if vd is not None else np.zeros(nv))
        vel[i] = vJ if p < 0 else K.Xup[i] @ vel[p] + vJ
        acc[i] = K.Xup[i] @ (a0 if p < 0 else acc[p]) + aJ + crm(vel[i]) @ vJ
        I = body_inertia(model, i)
        f[i] = I @ acc[i] + crf(vel[i]) @ (I @ vel[i])
        if fext is not None:
            f[i] = f[i] - np.linalg.inv(K.X0[i]).T @ fext[6 * i:6 * i + 6]
    tau = np.zeros(model.nv)
    for i in range(n - 1, -1, -1):
        p, vo, nv = int(model.parent[i]), int(model.v_offset[i]), NV[int(model.joint_type[i])]
        tau[vo:vo + nv] = K.S[i].T @ f[i]
        if p >= 0:
            f[p] = f[p] + K.Xup[i].T @ f[i]
    return tau


def crba(model, q):
    """RBDA Table 6.2 (composite rigid body algorithm), full symmetric H."""
    n = model.n_bodies
    K = Kin(model, q)
    Ic = [body_inertia(model, i) for i in range(n)]
    for i in range(n - 1, -1, -1):
        p = int(model.parent[i])
        if p >= 0:
            Ic[p] = Ic[p] + K.Xup[i].T @ Ic[i] @ K.Xup[i]
    H = np.zeros((model.nv, model.nv))
    for i in range(n):
        vo, nv = int(model.v_offset[i]), NV[int(model.joint_type[i])]
        if nv == 0:
            continue
        F = Ic[i] @ K.S[i]
        H[vo:vo + nv, vo:vo + nv] = K.S[i].T @ F
        j = i
        while int(model.parent[j]) >= 0:
            F = K.Xup[j].T @ F
            j = int(model.parent[j])
            wo, nw = int(model.v_offset[j]), NV[int(model.joint_type[j])]
            if nw:
                H[vo:vo + nv, wo:wo + nw] = F.T @ K.S[j]
                H[wo:wo + nw, vo:vo + nv] = (F.T @ K.S[j]).T
    return H


def aba(model, q, v, tau=None, fext=None):
    """RBDA Table 7.1 (articulated-body algorithm) in body coordinates."""
    n = model.n_bodies
    K = Kin(model, q)
    tau = np.zeros(model.nv) if tau is None else tau
    a0 = np.concatenate([np.zeros(3), -np.asarray(model.gravity, float)])
    vel, c, IA, pA = [None] * n, [None] * n, [None] * n, [None] * n
    for i in range(n):
        p, vo, nv = int(model.parent[i]), int(model.v_offset[i]), NV[int(model.joint_type[i])]
        vJ = K.S[i] @ v[vo:vo + nv]
        vel[i] = vJ if p < 0 else K.Xup[i] @ vel[p] + vJ
        c[i] = crm(vel[i]) @ vJ
        IA[i] = body_inertia(model, i)
        pA[i] = crf(vel[i]) @ (IA[i] @ vel[i])
        if fext is not None:
            pA[i] = pA[i] - np.linalg.inv(K.X0[i]).T @ fext[6 * i:6 * i + 6]
    U, Dinv, u = [None] * n, [None] * n, [None] * n
    for i in range(n - 1, -1, -1):
        p, vo, nv = int(model.parent[i]), int(model.v_offset[i]), NV[int(model.joint_type[i])]
        if nv:
            U[i] = IA[i] @ K.S[i]
            Dinv[i] = np.linalg.inv(K.S[i].T @ U[i])
            u[i] = tau[vo:vo + nv] - K.S[i].T @ pA[i]
            Ia = IA[i] - U[i] @ Dinv[i] @ U[i].T
            pa = pA[i] + Ia @ c[i] + U[i] @ Dinv[i] @ u[i]
        else:
            Ia, pa = IA[i], pA[i] + IA[i] @ c[i]
        if p >= 0:
            IA[p] = IA[p] + K.Xup[i].T @ Ia @ K.Xup[i]
            pA[p] = pA[p] + K.Xup[i].T @ pa
    vd = np.zeros(model.nv)
    a = [None] * n
    for i in range(n):
        p, vo, nv = int(model.parent[i]), int(model.v_offset[i]), NV[int(model.joint_type[i])]
        ap = K.Xup[i] @ (a0 if p < 0 else a[p]) + c[i]
        if nv:
            vd[vo:vo + nv] = Dinv[i] @ (u[i] - U[i].T @ ap)
            a[i] = ap + K.S[i] @ vd[vo:vo + nv]
        else:
            a[i] = ap
    return vd


def batch(fn, model, *arrays):
    """Apply a single-state function over the leading (batch) axis; None arguments stay None."""
    B = next(a.shape[0] for a in arrays if a is not None)
    return np.stack([fn(model, *[None if a is None else a[b] for a in arrays]) for b in range(B)])
