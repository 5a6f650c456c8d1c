"""ctypes front-end of oracle/librbd_oracle.so — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module. All batch
arrays are AOS: shape (B, n), one state per row (== one Julia column)."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

WHAT_DYNAMICS, WHAT_INVERSE_DYNAMICS, WHAT_DYNAMICS_BIAS, WHAT_MASS_MATRIX, WHAT_ABA = range(5)


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "librbd_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("rbd_oracle.c", "rbd_oracle_impl.h")] + [os.path.join(_HERE, "..", "include", "rbd_hip.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "librbd_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "librbd_oracle.so")
        if not os.path.exists(so):
            build()
        _LIB = ctypes.CDLL(so)
    return _LIB


def _sfx(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float64:
        return "_f64", ctypes.c_double
    if dtype == np.float32:
        return "_f32", ctypes.c_float
    raise TypeError(dtype)


def _ptr(a, ct):
    return None if a is None else a.ctypes.data_as(ctypes.POINTER(ct))


def _prep(a, dtype, shape):
    if a is None:
        return None
    a = np.ascontiguousarray(a, dtype=dtype)
    assert a.shape == shape, (a.shape, shape)
    return a


def batch(model, what, q, v=None, x=None, fext=None, dtype=np.float64, nthreads=1, want_qdot=False):
    """Run the oracle over a batch. q:(B,nq) v:(B,nv) x: tau (dynamics/ABA) or vdot (inverse dynamics) (B,nv),
    fext:(B, 6*n_bodies). Returns out (B,nv) [or (B,nv,nv) col-major-per-state for the mass matrix, i.e.
    out[b].T is M in row-major numpy indexing M[i,j]] and, when want_qdot, (out, qdot)."""
    sfx, ct = _sfx(dtype)
    B = q.shape[0]
    nq, nv, nb = model.nq, model.nv, model.n_bodies
    q = _prep(q, dtype, (B, nq))
    v = _prep(v, dtype, (B, nv))
    x = _prep(x, dtype, (B, nv))
    fext = _prep(fext, dtype, (B, 6 * nb))
    out = np.zeros((B, nv * nv) if what == WHAT_MASS_MATRIX else (B, nv), dtype=dtype)
    qdot = np.zeros((B, nq), dtype=dtype) if want_qdot else None
    f = getattr(lib(), "rbdo_batch" + sfx)
    f.restype = ctypes.c_int
    st = f(ctypes.byref(model.c_struct()), ctypes.c_int(what), ctypes.c_int(B), ctypes.c_int(nthreads),
           _ptr(q, ct), _ptr(v, ct), _ptr(x, ct), _ptr(fext, ct), _ptr(out, ct), _ptr(qdot, ct))
    if st != 0:
        raise RuntimeError(f"oracle status {st}")
    if what == WHAT_MASS_MATRIX:
        out = out.reshape(B, nv, nv).transpose(0, 2, 1)  # -> out[b, i, j] = M[i, j]
    return (out, qdot) if want_qdot else out


def dynamics(model, q, v, tau=None, fext=None, **kw):
    return batch(model, WHAT_DYNAMICS, q, v, tau, fext, **kw)


def aba(model, q, v, tau=None, fext=None, **kw):
    return batch(model, WHAT_ABA, q, v, tau, fext, **kw)


def inverse_dynamics(model, q, v, vdot, fext=None, **kw):
    return batch(model, WHAT_INVERSE_DYNAMICS, q, v, vdot, fext, **kw)


def dynamics_bias(model, q, v, fext=None, **kw):
    return batch(model, WHAT_DYNAMICS_BIAS, q, v, None, fext, **kw)


def mass_matrix(model, q, **kw):
    """Lower triangle valid (strict upper zero), like the reference's Symmetric(…, :L) storage."""
    return batch(model, WHAT_MASS_MATRIX, q, **kw)


def energy(model, q, v, dtype=np.float64):
    sfx, ct = _sfx(dtype)
    f = getattr(lib(), "rbdo_energy" + sfx)
    f.restype = ctypes.c_int
    B = q.shape[0]
    ke, pe = np.zeros(B, dtype), np.zeros(B, dtype)
    q = np.ascontiguousarray(q, dtype)
    v = np.ascontiguousarray(v, dtype)
    for b in range(B):
        k, p = ct(), ct()
        st = f(ctypes.byref(model.c_struct()), _ptr(q[b], ct), _ptr(v[b], ct), ctypes.byref(k), ctypes.byref(p))
        assert st == 0
        ke[b], pe[b] = k.value, p.value
    return ke, pe


def transforms(model, q, dtype=np.float64):
    """transforms_to_root per body: (B, n_bodies, 12) = R row-major (9) + p (3)."""
    sfx, ct = _sfx(dtype)
    f = getattr(lib(), "rbdo_transforms" + sfx)
    f.restype = ctypes.c_int
    B = q.shape[0]
    q = np.ascontiguousarray(q, dtype)
    out = np.zeros((B, model.n_bodies, 12), dtype)
    for b in range(B):
        assert f(ctypes.byref(model.c_struct()), _ptr(q[b], ct), _ptr(out[b], ct)) == 0
    return out


def dynamics_loops(model, q, v, tau=None, fext=None, stabilize=True, dtype=np.float64):
    """dynamics! for mechanisms with loop joints. Returns dict(vdot, qdot, lam, K [B,nc,nv], k, M [B,nv,nv] lower, c)."""
    sfx, ct = _sfx(dtype)
    f = getattr(lib(), "rbdo_dynamics_loops" + sfx)
    f.restype = ctypes.c_int
    B, nq, nv, nb, nc = q.shape[0], model.nq, model.nv, model.n_bodies, model.nc
    q = _prep(q, dtype, (B, nq)); v = _prep(v, dtype, (B, nv)); tau = _prep(tau, dtype, (B, nv)); fext = _prep(fext, dtype, (B, 6 * nb))
    out = dict(vdot=np.zeros((B, nv), dtype), qdot=np.zeros((B, nq), dtype), lam=np.zeros((B, max(nc, 1)), dtype),
               K=np.zeros((B, nv, max(nc, 1)), dtype), k=np.zeros((B, max(nc, 1)), dtype), M=np.zeros((B, nv, nv), dtype), c=np.zeros((B, nv), dtype))
    for b in range(B):
        st = f(ctypes.byref(model.c_struct()), _ptr(q[b], ct), _ptr(v[b], ct), _ptr(None if tau is None else tau[b], ct),
               _ptr(None if fext is None else fext[b], ct), ctypes.c_int(1 if stabilize else 0), _ptr(out["vdot"][b], ct), _ptr(out["qdot"][b], ct),
               _ptr(out["lam"][b], ct), _ptr(out["K"][b], ct), _ptr(out["k"][b], ct), _ptr(out["M"][b], ct), _ptr(out["c"][b], ct))
        if st != 0:
            raise RuntimeError(f"oracle status {st}")
    out["K"] = out["K"].transpose(0, 2, 1)[:, :nc]   # column-major (nv, nc) per state -> [b, c, v]
    out["M"] = out["M"].transpose(0, 2, 1)
    out["lam"], out["k"] = out["lam"][:, :nc], out["k"][:, :nc]
    return out


def momentum_matrix(model, q, v=None, dtype=np.float64):
    """(A [B, 6, nv], hsum [B, 6] = Σ_b I_b T_b (when v is given), com [B, 3])."""
    sfx, ct = _sfx(dtype)
    f = getattr(lib(), "rbdo_momentum_matrix" + sfx)
    f.restype = ctypes.c_int
    B = q.shape[0]
    q = np.ascontiguousarray(q, dtype)
    v = None if v is None else np.ascontiguousarray(v, dtype)
    A = np.zeros((B, model.nv, 6), dtype); h = np.zeros((B, 6), dtype); com = np.zeros((B, 3), dtype)
    for b in range(B):
        assert f(ctypes.byref(model.c_struct()), _ptr(q[b], ct), _ptr(None if v is None else v[b], ct), _ptr(A[b], ct), _ptr(h[b], ct), _ptr(com[b], ct)) == 0
    return A.transpose(0, 2, 1), h, com


def geometric_jacobian(model, q, base, body, v=None, dtype=np.float64):
    """(J [B, 6, nv] of path(base -> body) in the root frame, relative twist [B, 6] when v is given)."""
    sfx, ct = _sfx(dtype)
    f = getattr(lib(), "rbdo_geometric_jacobian" + sfx)
    f.restype = ctypes.c_int
    B = q.shape[0]
    q = np.ascontiguousarray(q, dtype)
    v = None if v is None else np.ascontiguousarray(v, dtype)
    J = np.zeros((B, model.nv, 6), dtype); t = np.zeros((B, 6), dtype)
    for b in range(B):
        assert f(ctypes.byref(model.c_struct()), _ptr(q[b], ct), _ptr(None if v is None else v[b], ct), int(base), int(body), _ptr(J[b], ct), _ptr(t[b], ct)) == 0
    return J.transpose(0, 2, 1), t


def body_kinematics(model, q, v, vdot, dtype=np.float64):
    """Per body: transform_to_root (B, nb, 12: R row-major, p), twist_wrt_world (B, nb, 6) and spatial acceleration relative to
    the root (B, nb, 6), all in the root frame."""
    sfx, ct = _sfx(dtype)
    f = getattr(lib(), "rbdo_body_kinematics" + sfx)
    f.restype = ctypes.c_int
    B = q.shape[0]
    q, v, vdot = (np.ascontiguousarray(a, dtype) for a in (q, v, vdot))
    H = np.zeros((B, model.n_bodies, 12), dtype); T = np.zeros((B, model.n_bodies, 6), dtype); A = np.zeros((B, model.n_bodies, 6), dtype)
    for b in range(B):
        assert f(ctypes.byref(model.c_struct()), _ptr(q[b], ct), _ptr(v[b], ct), _ptr(vdot[b], ct), _ptr(H[b], ct), _ptr(T[b], ct), _ptr(A[b], ct)) == 0
    return H, T, A


def inverse_dynamics_bodies(model, q, v, vdot=None, fext=None, dtype=np.float64):
    """(tau [B, nv], jointwrenches [B, nb, 6], accelerations [B, nb, 6]) of inverse_dynamics! (vdot given) / dynamics_bias! (vdot None),
    root frame; accelerations include the root's −gravity as the reference's spatial_accelerations! / bias_accelerations! leave them."""
    sfx, ct = _sfx(dtype)
    f = getattr(lib(), "rbdo_inverse_dynamics_bodies" + sfx)
    f.restype = ctypes.c_int
    B, nb = q.shape[0], model.n_bodies
    q, v = np.ascontiguousarray(q, dtype), np.ascontiguousarray(v, dtype)
    vdot = None if vdot is None else np.ascontiguousarray(vdot, dtype)
    fext = None if fext is None else np.ascontiguousarray(fext, dtype)
    tau, jw, acc = np.zeros((B, model.nv), dtype), np.zeros((B, nb, 6), dtype), np.zeros((B, nb, 6), dtype)
    for b in range(B):
        assert f(ctypes.byref(model.c_struct()), _ptr(q[b], ct), _ptr(v[b], ct), _ptr(None if vdot is None else vdot[b], ct),
                 _ptr(None if fext is None else fext[b], ct), _ptr(tau[b], ct), _ptr(jw[b], ct), _ptr(acc[b], ct)) == 0
    return tau, jw, acc


def momentum(model, q, v, dtype=np.float64):
    """(momentum [B, 6], momentum_rate_bias [B, 6]) in the root frame."""
    sfx, ct = _sfx(dtype)
    f = getattr(lib(), "rbdo_momentum" + sfx)
    f.restype = ctypes.c_int
    B = q.shape[0]
    q, v = np.ascontiguousarray(q, dtype), np.ascontiguousarray(v, dtype)
    h, hb = np.zeros((B, 6), dtype), np.zeros((B, 6), dtype)
    for b in range(B):
        assert f(ctypes.byref(model.c_struct()), _ptr(q[b], ct), _ptr(v[b], ct), _ptr(h[b], ct), _ptr(hb[b], ct)) == 0
    return h, hb


def contact_dynamics(model, q, v, s, dtype=np.float64):
    """contact_dynamics!(result, state): returns (s after the resets [B, ns], sdot [B, ns], contactwrenches [B, 6 n_bodies])."""
    sfx, ct = _sfx(dtype)
    f = getattr(lib(), "rbdo_contact_dynamics" + sfx)
    f.restype = ctypes.c_int
    B, ns = q.shape[0], model.ns
    q, v = np.ascontiguousarray(q, dtype), np.ascontiguousarray(v, dtype)
    s = np.array(s, dtype).reshape(B, ns).copy()
    sd, cw = np.zeros((B, ns), dtype), np.zeros((B, 6 * model.n_bodies), dtype)
    for b in range(B):
        assert f(ctypes.byref(model.c_struct()), _ptr(q[b], ct), _ptr(v[b], ct), _ptr(s[b], ct), _ptr(sd[b], ct), _ptr(cw[b], ct)) == 0
    return s, sd, cw


def dynamics_contact(model, q, v, s, tau=None, fext=None, dtype=np.float64):
    """dynamics!(result, state, τ, wext) with contact points (mechanism_algorithms.jl:845-864): (vdot, s after resets, sdot, contactwrenches, totalwrenches)."""
    s2, sd, cw = contact_dynamics(model, q, v, s, dtype)
    tw = cw if fext is None else cw + np.asarray(fext, dtype)
    return dynamics(model, q, v, tau, tw, dtype=dtype), s2, sd, cw, tw
