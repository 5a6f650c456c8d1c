"""numpy prototype of the arithmetic of aba_track_kernel (rbd_track.hpp), checked against the oracle on the CPU.

Not product code and not the oracle: a scratch restatement of the *formulation* the track kernel uses, so that its algebra
(canonical body frames with the joint axis on z, bias accelerations folded into the bias force, 6-dof root solved in place)
is validated before any GPU time is spent.  Run: python scripts/proto_track.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

REV, PRIS, FLOAT, FIXED = 1, 2, 3, 0


def frame_with_z(axis):
    """Rotation P with P e_z = axis (signed permutation when the axis is a coordinate axis)."""
    a = np.asarray(axis, float)
    k = int(np.argmin(np.abs(a)))  # least aligned coordinate axis
    e = np.zeros(3); e[k] = 1.0
    x = e - a * (a @ e)
    x /= np.linalg.norm(x)
    y = np.cross(a, x)
    return np.stack([x, y, a], axis=1)


def skew(c):
    return np.array([[0, -c[2], c[1]], [c[2], 0, -c[0]], [-c[1], c[0], 0]])


def quat_R(w, x, y, z):
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def mcross(a, b):  # se3 commutator [a, b]
    return np.concatenate([np.cross(a[:3], b[:3]), np.cross(a[:3], b[3:]) + np.cross(a[3:], b[:3])])


def canonical(model):
    """Per body: C = P_p' Xpred_R P_b, pp = P_p' Xpred_p, J' = P_b' J P_b, mc' = P_b' mc (body frames re-based so every 1-dof axis is +z)."""
    n = model.n_bodies
    P = [np.eye(3)] * n
    for i in range(n):
        t = int(model.joint_type[i])
        if t in (REV, PRIS):
            P[i] = frame_with_z(model.joint_axis[i])
    rec = []
    for i in range(n):
        p = int(model.parent[i])
        Pp = np.eye(3) if p < 0 else P[p]
        rec.append(dict(C=Pp.T @ model.pred_rot[i] @ P[i], pp=Pp.T @ model.pred_trans[i], J=P[i].T @ model.inertia_moment[i] @ P[i],
                        mc=P[i].T @ model.inertia_cross[i], m=float(model.inertia_mass[i])))
    return rec


def aba_track(model, q, v, tau, fext):
    n = model.n_bodies
    rec = canonical(model)
    g = np.asarray(model.gravity, float)
    a0 = np.concatenate([np.zeros(3), -g])
    R = [None] * n; p = [None] * n; T = [None] * n; av = [None] * n; S = [None] * n
    IA = [None] * n; pA = [None] * n
    for i in range(n):
        par = int(model.parent[i]); t = int(model.joint_type[i]); r = rec[i]
        pR, pp, pT, pav = (np.eye(3), np.zeros(3), np.zeros(6), a0) if par < 0 else (R[par], p[par], T[par], av[par])
        qo, vo = int(model.q_offset[i]), int(model.v_offset[i])
        if t == FLOAT:
            assert par < 0
            qq = q[qo:qo + 7]
            R[i] = pR @ r["C"] @ quat_R(*qq[:4])
            p[i] = pp + pR @ (r["pp"] + r["C"] @ qq[4:7])
            w = R[i] @ v[vo:vo + 3]
            T[i] = np.concatenate([w, R[i] @ v[vo + 3:vo + 6] + np.cross(p[i], w)])
            av[i] = pav.copy()  # [T, T] = 0
            S[i] = None
        else:
            M = pR @ r["C"]
            if t == REV:
                s, c = np.sin(q[qo]), np.cos(q[qo])
                R[i] = np.stack([c * M[:, 0] + s * M[:, 1], c * M[:, 1] - s * M[:, 0], M[:, 2]], axis=1)
                p[i] = pp + pR @ r["pp"]
                z = R[i][:, 2]
                S[i] = np.concatenate([z, np.cross(p[i], z)])
            elif t == PRIS:
                R[i] = M
                z = M[:, 2]
                p[i] = pp + pR @ r["pp"] + z * q[qo]
                S[i] = np.concatenate([np.zeros(3), z])
            elif t == FIXED:
                R[i] = M
                p[i] = pp + pR @ r["pp"]
                S[i] = np.zeros(6)
            else:
                raise NotImplementedError
            vJ = S[i] * (v[vo] if t != FIXED else 0.0)
            T[i] = pT + vJ
            av[i] = pav + mcross(pT, vJ)
        # inertia to root
        Rm, J, mc, m = R[i], r["J"], r["mc"], r["m"]
        Rmc = Rm @ mc
        c = Rmc + m * p[i]
        Jw = Rm @ J @ Rm.T
        Y = np.outer(Rmc, p[i]) + np.outer(p[i], c)
        Jw = Jw - Y + np.trace(Y) * np.eye(3)
        I6 = np.zeros((6, 6))
        I6[:3, :3] = Jw; I6[:3, 3:] = skew(c); I6[3:, :3] = skew(c).T; I6[3:, 3:] = m * np.eye(3)
        IA[i] = I6
        h = I6 @ T[i]
        w_, v_ = T[i][:3], T[i][3:]
        pA[i] = I6 @ av[i] + np.concatenate([np.cross(w_, h[:3]) + np.cross(v_, h[3:]), np.cross(w_, h[3:])]) - fext[6 * i:6 * i + 6]
    W = [None] * n; ud = [None] * n; ad = [None] * n
    vd = np.zeros(model.nv)
    for i in range(n - 1, -1, -1):
        par = int(model.parent[i]); t = int(model.joint_type[i]); vo = int(model.v_offset[i])
        if t == FLOAT:
            Xf_tau = np.concatenate([R[i] @ tau[vo:vo + 3] + np.cross(p[i], R[i] @ tau[vo + 3:vo + 6]), R[i] @ tau[vo + 3:vo + 6]])
            ad[i] = np.linalg.solve(IA[i], Xf_tau - pA[i])
            d = ad[i]
            vd[vo:vo + 3] = R[i].T @ d[:3]
            vd[vo + 3:vo + 6] = R[i].T @ (d[3:] - np.cross(p[i], d[:3]))
            continue
        if t == FIXED:
            U = np.zeros(6); W[i] = np.zeros(6); ud[i] = 0.0
        else:
            U = IA[i] @ S[i]
            Dinv = 1.0 / (S[i] @ U)
            u = tau[vo] - S[i] @ pA[i]
            W[i] = U * Dinv; ud[i] = u * Dinv
        if par >= 0:
            IA[par] = IA[par] + IA[i] - np.outer(W[i], U)
            pA[par] = pA[par] + pA[i] + U * ud[i]
    for i in range(n):
        par = int(model.parent[i]); t = int(model.joint_type[i]); vo = int(model.v_offset[i])
        if t == FLOAT:
            continue
        ap = np.zeros(6) if par < 0 else ad[par]
        if t == FIXED:
            ad[i] = ap
            continue
        vd[vo] = ud[i] - W[i] @ ap
        ad[i] = ap + S[i] * vd[vo]
    return vd


def main():
    import conftest  # noqa: F401  (path set-up)
    import rbd_amd as rbd
    import oracle
    oracle.build()
    models = {name: rbd.load_flat_model(os.path.join(ROOT, "tests", "golden", "models", name + ".json"))
              for name in ("atlas_floating", "atlas_fixed", "acrobot_urdf", "valkyrie_floating")}
    models["double_pendulum"] = rbd.flatten(rbd.double_pendulum())
    worst = 0.0
    for name, model in models.items():
        rng = np.random.default_rng(3)
        B = 4
        q = rbd.rand_configuration(model, B, rng); v = rbd.rand_velocity(model, B, rng)
        tau = rng.random((B, model.nv)); fe = rng.random((B, 6 * model.n_bodies))
        ref = oracle.dynamics(model, q, v, tau, fe)
        for b in range(B):
            got = aba_track(model, q[b], v[b], tau[b], fe[b])
            err = np.abs(got - ref[b]).max() / max(1.0, np.abs(ref[b]).max())
            worst = max(worst, err)
        print(f"{name:20s} rel err {err:.3e}")
    print("worst", worst)
    assert worst < 1e-10


if __name__ == "__main__":
    main()
