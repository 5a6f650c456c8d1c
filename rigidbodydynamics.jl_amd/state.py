"""Batched mirror of the reference's operator interface for the hot path.

    reference (one state, CPU)                                this package (B states, one MI355X)
    MechanismState(mechanism)            mechanism_state.jl:79    MechanismState(mechanism, batch, dtype, device)
    DynamicsResult(mechanism)            dynamics_result.jl:37    DynamicsResult(mechanism, batch, dtype, device)
    dynamics!(result, state, τ, wext)    mechanism_algorithms.jl:845   dynamics_(result, state, torques, externalwrenches)
    inverse_dynamics!(τ, jw, acc, state, v̇, wext)      :542           inverse_dynamics_(torquesout, state, vd, externalwrenches)
    dynamics_bias!(result, state)                        :496           dynamics_bias_(result, state, externalwrenches)
    mass_matrix!(M, state) / (result, state)             :248/:274      mass_matrix_(M_or_result, state)

`q`, `v`, … are torch tensors on the GPU with ONE STATE PER COLUMN of the Julia matrix, i.e. shape (B, n) row-major
(== Julia `n × B` column-major, RBD_LAYOUT_AOS) by default, or (n, B) with layout="soa".  Julia's `!` becomes a
trailing underscore.  All arithmetic happens in csrc/librbd_hip.so; errors map to the reference's exception types.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import numpy as np
import torch

from . import _capi
from .mechanism import FlatModel, Mechanism, flatten


class DimensionMismatch(ValueError):
    """Julia's DimensionMismatch (e.g. src/mechanism_algorithms.jl:250)."""


def _raise(status, where):
    try:
        _capi.check(status, where)
    except _capi.RBDError as e:
        if e.status == 2:
            raise DimensionMismatch(str(e)) from None
        if e.status == 1:
            raise ValueError(str(e)) from None  # ArgumentError
        if e.status == 7:
            raise RuntimeError("This method can currently only handle tree Mechanisms.") from None
        raise


class _Model:
    """Owns the rbd_model_t handle for a FlatModel (re-flatten when `modcount` changes, like @modcountcheck)."""

    def __init__(self, flat: FlatModel):
        self.flat = flat
        h = ctypes.c_void_p()
        _raise(_capi.lib().rbd_model_create(ctypes.cast(ctypes.byref(flat.c_struct()), ctypes.c_void_p), ctypes.byref(h)), "rbd_model_create")
        self.handle = h

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                _capi.lib().rbd_model_destroy(self.handle)
        except Exception:
            pass


def _as_model(mechanism_or_flat) -> _Model:
    if isinstance(mechanism_or_flat, _Model):
        return mechanism_or_flat
    if isinstance(mechanism_or_flat, Mechanism):
        cached = getattr(mechanism_or_flat, "_rbd_model", None)
        if cached is None or cached[0] != mechanism_or_flat.modcount:
            cached = (mechanism_or_flat.modcount, _Model(flatten(mechanism_or_flat)))
            mechanism_or_flat._rbd_model = cached
        return cached[1]
    if isinstance(mechanism_or_flat, FlatModel):
        cached = getattr(mechanism_or_flat, "_rbd_model", None)
        if cached is None:
            cached = _Model(mechanism_or_flat)
            mechanism_or_flat._rbd_model = cached
        return cached
    raise TypeError(type(mechanism_or_flat))


_TORCH_DTYPE = {torch.float64: _capi.F64, torch.float32: _capi.F32}


class _Workspace:
    def __init__(self, model: _Model, batch: int, dtype: torch.dtype, device: torch.device):
        if device.type != "cuda":
            raise RuntimeError("rigidbodydynamics.jl_amd runs on an MI355X only (no CPU path); got device " + str(device))
        self.model, self.batch, self.dtype, self.device = model, batch, dtype, device
        h = ctypes.c_void_p()
        idx = device.index if device.index is not None else torch.cuda.current_device()
        stream = torch.cuda.current_stream(device).cuda_stream
        _raise(_capi.lib().rbd_workspace_create(model.handle, batch, idx, _TORCH_DTYPE[dtype], ctypes.c_void_p(stream), ctypes.byref(h)),
               "rbd_workspace_create")
        self.handle = h

    def use_current_stream(self):
        _capi.lib().rbd_workspace_set_stream(self.handle, ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                _capi.lib().rbd_workspace_destroy(self.handle)
        except Exception:
            pass


class MechanismState:
    """Batch of B states (q, v) of one mechanism on one GPU.  `layout="aos"`: q is (B, nq) — one state per Julia
    column; `layout="soa"`: q is (nq, B)."""

    def __init__(self, mechanism, batch: int = 1, dtype: torch.dtype = torch.float64, device="cuda", layout: str = "aos"):
        self.model = _as_model(mechanism)
        self.flat = self.model.flat
        self.batch, self.dtype, self.device, self.layout = int(batch), dtype, torch.device(device), layout
        if layout not in ("aos", "soa"):
            raise ValueError("layout must be 'aos' or 'soa'")
        self.ws = _Workspace(self.model, self.batch, dtype, self.device)
        self.q = self._zeros(self.flat.nq)
        self.v = self._zeros(self.flat.nv)
        self.s = self._zeros(getattr(self.flat, "ns", 0))  # additional states: the soft-contact friction states (mechanism_state.jl:64, :139-152)
        zero_configuration_(self)

    def _zeros(self, n):
        shape = (self.batch, n) if self.layout == "aos" else (n, self.batch)
        return torch.zeros(shape, dtype=self.dtype, device=self.device)

    def _opts(self, algorithm=_capi.ALGO_ABA, stabilization=1):
        return _capi.Opts(_capi.LAYOUT_AOS if self.layout == "aos" else _capi.LAYOUT_SOA, _capi.MEM_DEVICE, algorithm, stabilization)

    def _check(self, t: Optional[torch.Tensor], n: int, what: str):
        if t is None:
            return
        shape = (self.batch, n) if self.layout == "aos" else (n, self.batch)
        if tuple(t.shape) != shape:
            raise DimensionMismatch(f"{what}: expected shape {shape}, got {tuple(t.shape)}")
        if t.dtype != self.dtype or t.device.type != "cuda" or not t.is_contiguous():
            raise ValueError(f"{what}: must be a contiguous {self.dtype} CUDA tensor")

    def num_positions(self):
        return self.flat.nq

    def num_velocities(self):
        return self.flat.nv

    def num_additional_states(self):
        return getattr(self.flat, "ns", 0)


def _from_host(state: MechanismState, a: np.ndarray) -> torch.Tensor:
    t = torch.as_tensor(np.ascontiguousarray(a), dtype=state.dtype)
    if state.layout == "soa":
        t = t.t().contiguous()
    return t.to(state.device)


def set_configuration_(state: MechanismState, q):
    """`set_configuration!(state, q)`: q is (B, nq) host/torch data, one state per row."""
    q = q if isinstance(q, torch.Tensor) else _from_host(state, np.asarray(q))
    state._check(q, state.flat.nq, "q")
    state.q.copy_(q)


def set_velocity_(state: MechanismState, v):
    v = v if isinstance(v, torch.Tensor) else _from_host(state, np.asarray(v))
    state._check(v, state.flat.nv, "v")
    state.v.copy_(v)


def zero_configuration_(state: MechanismState):
    """`zero_configuration!`: identity quaternions (quaternion_floating.jl:169-173), sin/cos = (0, 1), zeros elsewhere."""
    from .mechanism import JOINT_QUAT_FLOATING, JOINT_QUAT_SPHERICAL, JOINT_SINCOS_REVOLUTE
    q = np.zeros((state.batch, state.flat.nq))
    for i in range(state.flat.n_bodies):
        t, o = int(state.flat.joint_type[i]), int(state.flat.q_offset[i])
        if t in (JOINT_QUAT_FLOATING, JOINT_QUAT_SPHERICAL):
            q[:, o] = 1.0
        elif t == JOINT_SINCOS_REVOLUTE:
            q[:, o + 1] = 1.0
    state.q.copy_(_from_host(state, q))
    state.v.zero_()


def rand_(state: MechanismState, seed: int = 0):
    """`rand!(state)` with the reference's per-joint distributions (SURVEY.md §8 d)."""
    from .mechanism import rand_configuration, rand_velocity
    rng = np.random.default_rng(seed)
    set_configuration_(state, rand_configuration(state.flat, state.batch, rng))
    set_velocity_(state, rand_velocity(state.flat, state.batch, rng))


class DynamicsResult:
    """Output container with the reference's field names (src/dynamics_result.jl:11-36), batched."""

    def __init__(self, mechanism, batch: int = 1, dtype: torch.dtype = torch.float64, device="cuda", layout: str = "aos", bodies: bool = False):
        """bodies=True also allocates the per-body fields `accelerations`, `jointwrenches`, `totalwrenches` (src/dynamics_result.jl:26-29;
        (B, 6*n_bodies) each, root frame) and makes `dynamics_` / `dynamics_bias_` fill them like the reference does."""
        self.model = _as_model(mechanism)
        f = self.model.flat
        self.batch, self.dtype, self.device, self.layout = int(batch), dtype, torch.device(device), layout
        z = lambda n: torch.zeros((batch, n) if layout == "aos" else (n, batch), dtype=dtype, device=self.device)
        self.accelerations = z(6 * f.n_bodies) if bodies else None   # bias accelerations after dynamics! / dynamics_bias! (mechanism_algorithms.jl:377-385)
        self.jointwrenches = z(6 * f.n_bodies) if bodies else None   # joint wrenches of the bias run (:442-459)
        self.totalwrenches = z(6 * f.n_bodies) if bodies else None   # external + contact wrenches (:851-855)
        self.massmatrix = z(f.nv * f.nv)       # nv×nv column-major per state, lower triangle valid (Symmetric 'L')
        self.dynamicsbias = z(f.nv)
        self.qd = z(f.nq)                        # q̇
        self.vd = z(f.nv)                        # v̇
        self.lambda_ = z(max(f.nc, 1))[:, :f.nc] if layout == "aos" else z(max(f.nc, 1))[:f.nc]  # λ
        ns = getattr(f, "ns", 0)
        self.sd = z(ns)                                               # ṡ (dynamics_result.jl:20)
        self.contactwrenches = z(6 * f.n_bodies) if ns > 0 else None  # (dynamics_result.jl:25), root frame
        if ns > 0 and self.totalwrenches is None:
            self.totalwrenches = z(6 * f.n_bodies)
        self.constraintjacobian = z(f.nc * f.nv)
        self.constraintbias = z(f.nc)

    def massmatrix_dense(self) -> torch.Tensor:
        """(B, nv, nv) symmetric matrices M[b, i, j] from the lower-triangular column-major storage."""
        nv = self.model.flat.nv
        m = self.massmatrix if self.layout == "aos" else self.massmatrix.t()
        L = torch.tril(m.reshape(self.batch, nv, nv).transpose(1, 2))
        return L + torch.tril(L, -1).transpose(1, 2)


def _ptr(t: Optional[torch.Tensor]):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _check_result(result: DynamicsResult, state: MechanismState):
    """A result built for another batch size, dtype, layout or device would be written out of bounds (the reference raises DimensionMismatch)."""
    if result.batch != state.batch or result.dtype != state.dtype or result.layout != state.layout:
        raise DimensionMismatch(f"DynamicsResult(batch={result.batch}, dtype={result.dtype}, layout={result.layout}) does not match "
                                f"MechanismState(batch={state.batch}, dtype={state.dtype}, layout={state.layout})")
    f = state.flat
    for t, n, name in ((result.vd, f.nv, "v̇"), (result.qd, f.nq, "q̇"), (result.dynamicsbias, f.nv, "dynamicsbias"), (result.massmatrix, f.nv * f.nv, "massmatrix"),
                       (result.accelerations, 6 * f.n_bodies, "accelerations"), (result.jointwrenches, 6 * f.n_bodies, "jointwrenches"),
                       (result.totalwrenches, 6 * f.n_bodies, "totalwrenches")):
        state._check(t, n, name)
    if f.nc > 0:
        state._check(result.constraintjacobian, f.nc * f.nv, "constraintjacobian")
        state._check(result.constraintbias, f.nc, "constraintbias")


class PDGains:
    """`PDGains(k, d)` — src/pdcontrol.jl:17-26."""

    def __init__(self, k: float, d: float):
        self.k, self.d = float(k), float(d)


class SE3PDGains:
    """`SE3PDGains(angular::PDGains, linear::PDGains)` — src/pdcontrol.jl:35-44; the value type of `stabilization_gains`
    (src/mechanism_algorithms.jl:614-626).  Scalar gains only (the reference also admits matrix gains in a frame; `default_constraint_stabilization_gains`
    and every use in the reference's tests are scalar)."""

    def __init__(self, angular: PDGains, linear: PDGains):
        if not isinstance(angular, PDGains) or not isinstance(linear, PDGains):
            raise ValueError("SE3PDGains(angular::PDGains, linear::PDGains)")
        self.angular, self.linear = angular, linear

    def as_tuple(self):
        return (self.angular.k, self.angular.d, self.linear.k, self.linear.d)


def default_constraint_stabilization_gains():
    """`default_constraint_stabilization_gains(T)` (src/mechanism_algorithms.jl:610-612): critical damping, T_stab = 0.1."""
    return SE3PDGains(PDGains(100.0, 20.0), PDGains(100.0, 20.0))


def _set_stabilization_gains(state: "MechanismState", stabilization_gains) -> int:
    """The `stabilization_gains` keyword of `dynamics!` / `simulate` (src/mechanism_algorithms.jl:614-632, :848; src/simulate.jl:37) → the flag of
    rbd_opts_t plus rbd_workspace_set_loop_gains.  Accepted, as in the reference: `None` (no stabilization); "default" (the model's gains — the
    reference's default_constraint_stabilization_gains unless the flat model says otherwise); one `SE3PDGains` for every loop joint (the ConstDict
    case); a dict from loop joint (name, or index in the model's loop-joint order) to `SE3PDGains` — every loop joint must have an entry, as
    `stabilization_gains[nontreejointid]` (:655) would throw a KeyError otherwise.  Anything else raises ValueError (Julia: a MethodError / ArgumentError)."""
    f = state.flat
    if stabilization_gains is None:
        return 0
    n = getattr(f, "n_loops", 0)
    if isinstance(stabilization_gains, str):
        if stabilization_gains != "default":
            raise ValueError(f"stabilization_gains: {stabilization_gains!r} (expected None, 'default', an SE3PDGains or a dict of them)")
        arr = None
    elif isinstance(stabilization_gains, SE3PDGains):
        arr = list(stabilization_gains.as_tuple()) * n
    elif isinstance(stabilization_gains, dict):
        arr = []
        for i, l in enumerate(f.loops):
            key = l.get("name") if l.get("name") in stabilization_gains else i
            if key not in stabilization_gains:
                raise KeyError(f"stabilization_gains has no entry for loop joint {l.get('name', i)!r}")
            g = stabilization_gains[key]
            if not isinstance(g, SE3PDGains):
                raise ValueError("stabilization_gains values must be SE3PDGains")
            arr += list(g.as_tuple())
    else:
        raise ValueError(f"stabilization_gains: unsupported {type(stabilization_gains).__name__} (expected None, 'default', an SE3PDGains or a dict of them)")
    if n:
        buf = None if arr is None else (ctypes.c_double * len(arr))(*arr)
        _raise(_capi.lib().rbd_workspace_set_loop_gains(state.ws.handle, buf), "rbd_workspace_set_loop_gains")
    return 1


def dynamics_(result: DynamicsResult, state: MechanismState, torques: Optional[torch.Tensor] = None,
              externalwrenches: Optional[torch.Tensor] = None, stabilization_gains="default", algorithm: str = "aba"):
    """`dynamics!(result, state, torques, externalwrenches; stabilization_gains)` (src/mechanism_algorithms.jl:845-864):
    fills `result.vd` (v̇) and `result.qd` (q̇).  `torques` (B, nv) defaults to zeros; `externalwrenches` is a dense
    (B, 6*n_bodies) tensor of root-frame wrenches (torque; force) per moving body (None == NullDict).
    algorithm="aba": fused articulated-body kernel (mapping chosen by batch size; "aba_compiled" / "aba_pipe" / "aba_walk" / "aba_tracks" / "aba_lanes" / "aba_banks" / "aba_chains" force one);
    "crba": the reference's own CRBA + Cholesky route, which also fills result.massmatrix and result.dynamicsbias."""
    f = state.flat
    _check_result(result, state)
    state._check(torques, f.nv, "torques")
    state._check(externalwrenches, 6 * f.n_bodies, "externalwrenches")
    state.ws.use_current_stream()
    algo = {"aba": _capi.ALGO_ABA, "crba": _capi.ALGO_CRBA_CHOLESKY, "aba_lanes": _capi.ALGO_ABA_LANES,
            "aba_chains": _capi.ALGO_ABA_CHAINS, "aba_banks": _capi.ALGO_ABA_BANKS, "aba_tracks": _capi.ALGO_ABA_TRACKS, "aba_walk": _capi.ALGO_ABA_WALK, "aba_pipe": _capi.ALGO_ABA_PIPE,
            "aba_compiled": _capi.ALGO_ABA_COMPILED}[algorithm]
    opts = state._opts(algo, _set_stabilization_gains(state, stabilization_gains))
    lam = result.lambda_ if f.nc > 0 else None
    if getattr(f, "ns", 0) > 0:
        # a mechanism with contact points: contact_dynamics! first, totalwrenches = externalwrenches + contactwrenches (:849-856); state.s is
        # reset where a point is not in contact, result.sd / contactwrenches / totalwrenches are filled
        state._check(state.s, f.ns, "s")
        state._check(result.sd, f.ns, "ṡ")
        st = _capi.lib().rbd_dynamics_contact(state.ws.handle, state.batch, _ptr(state.q), _ptr(state.v), _ptr(state.s), _ptr(torques), _ptr(externalwrenches),
                                              _ptr(result.vd), _ptr(result.qd), _ptr(result.sd), _ptr(result.contactwrenches), _ptr(result.totalwrenches),
                                              ctypes.byref(opts))
        _raise(st, "rbd_dynamics_contact")
        _fill_result_fields(result, state, algo, result.totalwrenches, totalwrenches_done=True)
        return None
    bind = algo == _capi.ALGO_CRBA_CHOLESKY and f.nc == 0  # (the Python mirror always hands device buffers over)
    if bind:  # the reference's route fills result.massmatrix / dynamicsbias: written in place, no copy out of the workspace afterwards
        _capi.lib().rbd_workspace_bind_result(state.ws.handle, _ptr(result.massmatrix), _ptr(result.dynamicsbias))
    try:
        st = _capi.lib().rbd_dynamics(state.ws.handle, state.batch, _ptr(state.q), _ptr(state.v), _ptr(torques), _ptr(externalwrenches),
                                      _ptr(result.vd), _ptr(result.qd), _ptr(lam), ctypes.byref(opts))
        _raise(st, "rbd_dynamics")
        if bind:
            _raise(_capi.lib().rbd_dynamics_result(state.ws.handle, state.batch, _ptr(result.massmatrix), _ptr(result.dynamicsbias), None, None, ctypes.byref(opts)),
                   "rbd_dynamics_result")
    finally:
        if bind:
            _capi.lib().rbd_workspace_bind_result(state.ws.handle, None, None)
    if (algo == _capi.ALGO_CRBA_CHOLESKY or f.nc > 0) and not bind:  # mechanisms with loop joints always take the reference's CRBA route
        st = _capi.lib().rbd_dynamics_result(state.ws.handle, state.batch, _ptr(result.massmatrix), _ptr(result.dynamicsbias),
                                             _ptr(result.constraintjacobian if f.nc else None),
                                             _ptr(result.constraintbias if f.nc else None), ctypes.byref(opts))
        _raise(st, "rbd_dynamics_result")
    _fill_result_fields(result, state, None, externalwrenches)
    return None


def _fill_result_fields(result, state, algo, wrenches, totalwrenches_done=False):
    """What dynamics! leaves in a DynamicsResult beside v̇: totalwrenches = externalwrenches + contact wrenches, and the bias accelerations /
    joint wrenches of its dynamics_bias! call with those total wrenches (src/mechanism_algorithms.jl:851-856); with algo (the contact path,
    where the forward dynamics ran as ABA whatever was asked for) the mass matrix of the "crba" route as well."""
    if result.accelerations is not None:
        if not totalwrenches_done:
            if wrenches is None:
                result.totalwrenches.zero_()
            else:
                result.totalwrenches.copy_(wrenches)
        o2 = state._opts()
        st = _capi.lib().rbd_dynamics_bias_bodies(state.ws.handle, state.batch, _ptr(state.q), _ptr(state.v), _ptr(wrenches), _ptr(result.dynamicsbias),
                                                  _ptr(result.jointwrenches), _ptr(result.accelerations), ctypes.byref(o2))
        _raise(st, "rbd_dynamics_bias_bodies")
    if algo == _capi.ALGO_CRBA_CHOLESKY:
        mass_matrix_(result, state)
        if result.accelerations is None:
            dynamics_bias_(result.dynamicsbias, state, wrenches)


_MAPPING = {"auto": _capi.ALGO_ABA, "lanes": _capi.ALGO_ABA_LANES, "banks": _capi.ALGO_ABA_BANKS, "walk": _capi.ALGO_ABA_WALK, "pipe": _capi.ALGO_ABA_PIPE,
            "compiled": _capi.ALGO_ABA_COMPILED}


def inverse_dynamics_(torquesout: torch.Tensor, state: MechanismState, vd: torch.Tensor,
                      externalwrenches: Optional[torch.Tensor] = None, mapping: str = "auto",
                      jointwrenchesout: Optional[torch.Tensor] = None, accelerations: Optional[torch.Tensor] = None):
    """`inverse_dynamics!(torquesout, jointwrenchesout, accelerations, state, v̇, externalwrenches)` (:542-553).  `jointwrenchesout` /
    `accelerations` (optional, (B, 6*n_bodies), root frame): the per-body outputs the reference fills — the wrench across the joint above each
    body and each body's spatial acceleration (root acceleration −gravity included, as `spatial_accelerations!` leaves it).
    mapping: lane mapping of the kernel ("auto": by batch size; "lanes" / "banks" / "walk" force one — tests, benchmarks)."""
    f = state.flat
    state._check(torquesout, f.nv, "torquesout")
    state._check(vd, f.nv, "v̇")
    state._check(externalwrenches, 6 * f.n_bodies, "externalwrenches")
    state._check(jointwrenchesout, 6 * f.n_bodies, "jointwrenchesout")
    state._check(accelerations, 6 * f.n_bodies, "accelerations")
    state.ws.use_current_stream()
    opts = state._opts(_MAPPING[mapping])
    if jointwrenchesout is None and accelerations is None:
        st = _capi.lib().rbd_inverse_dynamics(state.ws.handle, state.batch, _ptr(state.q), _ptr(state.v), _ptr(vd), _ptr(externalwrenches),
                                              _ptr(torquesout), ctypes.byref(opts))
    else:
        st = _capi.lib().rbd_inverse_dynamics_bodies(state.ws.handle, state.batch, _ptr(state.q), _ptr(state.v), _ptr(vd), _ptr(externalwrenches),
                                                     _ptr(torquesout), _ptr(jointwrenchesout), _ptr(accelerations), ctypes.byref(opts))
    _raise(st, "rbd_inverse_dynamics")
    return torquesout


def dynamics_bias_(result_or_out, state: MechanismState, externalwrenches: Optional[torch.Tensor] = None, mapping: str = "auto"):
    """`dynamics_bias!(result, state)` / `dynamics_bias!(torques, …, state, externalwrenches)` (:484-498)."""
    out = result_or_out.dynamicsbias if isinstance(result_or_out, DynamicsResult) else result_or_out
    f = state.flat
    state._check(out, f.nv, "dynamicsbias")
    state._check(externalwrenches, 6 * f.n_bodies, "externalwrenches")
    state.ws.use_current_stream()
    opts = state._opts(_MAPPING[mapping])
    if isinstance(result_or_out, DynamicsResult) and result_or_out.accelerations is not None:
        _check_result(result_or_out, state)
        st = _capi.lib().rbd_dynamics_bias_bodies(state.ws.handle, state.batch, _ptr(state.q), _ptr(state.v), _ptr(externalwrenches), _ptr(out),
                                                  _ptr(result_or_out.jointwrenches), _ptr(result_or_out.accelerations), ctypes.byref(opts))
    else:
        st = _capi.lib().rbd_dynamics_bias(state.ws.handle, state.batch, _ptr(state.q), _ptr(state.v), _ptr(externalwrenches), _ptr(out),
                                           ctypes.byref(opts))
    _raise(st, "rbd_dynamics_bias")
    return out


def mass_matrix_(M_or_result, state: MechanismState):
    """`mass_matrix!(M::Symmetric, state)` / `mass_matrix!(result, state)` (:248-274): M is (B, nv*nv), each row an
    nv×nv column-major matrix whose LOWER triangle is written (uplo == 'L')."""
    out = M_or_result.massmatrix if isinstance(M_or_result, DynamicsResult) else M_or_result
    f = state.flat
    state._check(out, f.nv * f.nv, "mass matrix has wrong size")
    state.ws.use_current_stream()
    opts = state._opts()
    st = _capi.lib().rbd_mass_matrix(state.ws.handle, state.batch, _ptr(state.q), _ptr(out), ctypes.byref(opts))
    _raise(st, "rbd_mass_matrix")
    return out


def mass_matrix_solve_(x: torch.Tensor, state: MechanismState, rhs: torch.Tensor, M_out: Optional[torch.Tensor] = None,
                       algorithm: str = "cholesky", packed: bool = False):
    """x = M(q)⁻¹ rhs.  algorithm="cholesky": CRBA + batched lower Cholesky — `dynamics_solve!`'s potrf!/potrs! branch (:764, :819);
    algorithm="aba": the O(n) articulated-body solve (same x, M never formed unless `M_out` is given).
    packed=True (cholesky): `M_out` is (B, nv (nv + 1) / 2) — LAPACK's packed lower triangle, element (i, j), i ≥ j, at i + j (2 nv − j − 1) / 2: the part of M
    the reference defines, half the bytes (`rbd_mass_matrix_solve_packed`; `unpack_lower` turns it into (B, nv, nv))."""
    f = state.flat
    state._check(x, f.nv, "x")
    state._check(rhs, f.nv, "rhs")
    if packed:
        if M_out is None or algorithm != "cholesky":
            raise ValueError("packed=True needs M_out and algorithm='cholesky'")
        state._check(M_out, f.nv * (f.nv + 1) // 2, "M_out (packed)")
        state.ws.use_current_stream()
        opts = state._opts(_capi.ALGO_CRBA_CHOLESKY)
        st = _capi.lib().rbd_mass_matrix_solve_packed(state.ws.handle, state.batch, _ptr(state.q), _ptr(rhs), _ptr(x), _ptr(M_out), ctypes.byref(opts))
        _raise(st, "rbd_mass_matrix_solve_packed")
        return x
    state._check(M_out, f.nv * f.nv, "M_out")
    state.ws.use_current_stream()
    opts = state._opts(_capi.ALGO_CRBA_CHOLESKY if algorithm == "cholesky" else _capi.ALGO_ABA)
    st = _capi.lib().rbd_mass_matrix_solve(state.ws.handle, state.batch, _ptr(state.q), _ptr(rhs), _ptr(x), _ptr(M_out), ctypes.byref(opts))
    _raise(st, "rbd_mass_matrix_solve")
    return x


def unpack_lower(packed, nv: int):
    """(B, nv (nv + 1) / 2) packed lower triangles (LAPACK 'L': columns back to back from their diagonals down) -> (B, nv, nv) with the strict upper part zero."""
    import numpy as np
    P = packed.detach().cpu().numpy() if hasattr(packed, "detach") else np.asarray(packed)
    out = np.zeros((P.shape[0], nv, nv), dtype=P.dtype)
    k = 0
    for j in range(nv):
        out[:, j:, j] = P[:, k:k + nv - j]
        k += nv - j
    return out


def sync(state: MechanismState) -> int:
    """`rbd_sync`: wait for the workspace's stream; returns the status (8 == some mass matrix was not positive definite,
    the batched analogue of LAPACK.potrf!'s PosDefException)."""
    state.ws.use_current_stream()
    return int(_capi.lib().rbd_sync(state.ws.handle))


def last_kernel(state: MechanismState) -> str:
    """Name(s) of the kernel(s) the last dynamics / solve call on this state's workspace was dispatched to (`rbd_workspace_last_kernel`)."""
    return (_capi.lib().rbd_workspace_last_kernel(state.ws.handle) or b"").decode()


RK4_C = (0.0, 0.5, 0.5, 1.0)  # c = row sums of the runge_kutta_4 tableau (src/ode_integrators.jl:48-55)


class TorqueTable:
    """Device-side open-loop controller for `simulate_`: `torques[k]` is the (B, nv) torque of entry k — entry 4·step + stage with
    `per_stage=True` (the stage times t, t + h/2, t + h/2, t + h at which the reference calls control!(τ, t, state), src/simulate.jl:42-48), entry
    `step` otherwise (zero-order hold).  No host round trip per stage (`rbd_simulate_controlled`, RBD_CONTROL_TABLE)."""

    def __init__(self, torques: torch.Tensor, per_stage: bool = True):
        self.torques, self.per_stage = torques, bool(per_stage)


class PDControl:
    """Device-side PD controller for `simulate_`: τ_i = torques_i − kp_i (q_i − q_des_i) − kd_i v_i on every Revolute / Prismatic joint, evaluated on the
    stage state inside the dynamics launch (`rbd_simulate_controlled`, RBD_CONTROL_PD).  kp, kd: (nv,) tensors; q_des: (B, nq) or None (zero)."""

    def __init__(self, kp: torch.Tensor, kd: torch.Tensor, q_des: Optional[torch.Tensor] = None, torques: Optional[torch.Tensor] = None):
        self.kp, self.kd, self.q_des, self.torques = kp, kd, q_des, torques


def simulate_(state: MechanismState, final_time: float, control_=None, dt: float = 1e-4, torques: Optional[torch.Tensor] = None,
              externalwrenches: Optional[torch.Tensor] = None, stabilization_gains="default", store: bool = False):
    """`simulate(state0, final_time, control!; Δt, stabilization_gains)` (src/simulate.jl:36-55) for a whole batch in lockstep:
    Munthe-Kaas RK4 on the device, `state.q` / `state.v` advanced in place.

    * `control_ is None`: constant `torques` (default zero, the reference's `zero_torque!`) — every stage of every step runs on
      the GPU without returning to the host (`rbd_simulate`).
    * `control_(torques, t, state)`: called before each stage's `dynamics!` exactly like the reference's closure
      (src/simulate.jl:42-48); it must fill the (B, nv) `torques` tensor in place.
    Returns `ts` (and the lists `qs`, `vs` of per-step snapshots when `store`), with the reference's loop `while t < final_time`."""
    f = state.flat
    state._check(torques, f.nv, "torques")
    state._check(externalwrenches, 6 * f.n_bodies, "externalwrenches")
    state.ws.use_current_stream()
    opts = state._opts(_capi.ALGO_ABA, _set_stabilization_gains(state, stabilization_gains))
    L = _capi.lib()
    ts, t = [0.0], 0.0
    qs, vs = ([state.q.clone()], [state.v.clone()]) if store else (None, None)
    nsteps = 0
    while t < final_time:  # ode_integrators.jl:311-314
        t += dt
        ts.append(t)
        nsteps += 1
    if getattr(f, "ns", 0) > 0:  # soft contact: the additional state is integrated beside (q, v)
        if control_ is not None:
            raise NotImplementedError("control callbacks with contact points: use dynamics_ per stage")
        ss = [state.s.clone()] if store else None
        for _ in range(nsteps if store else 1):
            _raise(L.rbd_simulate_contact(state.ws.handle, state.batch, _ptr(state.q), _ptr(state.v), _ptr(state.s), _ptr(torques), _ptr(externalwrenches),
                                          ctypes.c_double(dt), 1 if store else nsteps, ctypes.byref(opts)), "rbd_simulate_contact")
            if store:
                qs.append(state.q.clone()); vs.append(state.v.clone()); ss.append(state.s.clone())
        return (np.array(ts), qs, vs, ss) if store else np.array(ts)
    if isinstance(control_, (TorqueTable, PDControl)):
        if torques is not None:
            raise ValueError("pass the torques through the controller object")
        ctl = _capi.Control()
        if isinstance(control_, TorqueTable):
            per = 4 if control_.per_stage else 1
            tt = control_.torques
            if tt.dtype != state.dtype or not tt.is_cuda or not tt.is_contiguous() or tt.dim() != 3 or tt.shape[0] < per * nsteps:
                raise DimensionMismatch(f"TorqueTable needs a contiguous device tensor of at least {per * nsteps} entries of the batch's (B, nv) shape")
            state._check(tt[0], f.nv, "torque table entry")
            ctl.kind, ctl.per_stage, ctl.tau = _capi.CONTROL_TABLE, int(control_.per_stage), tt.data_ptr()
            keep = (tt,)
        else:
            kp = control_.kp.to(device=state.device, dtype=state.dtype).contiguous()
            kd = control_.kd.to(device=state.device, dtype=state.dtype).contiguous()
            if tuple(kp.shape) != (f.nv,) or tuple(kd.shape) != (f.nv,):
                raise DimensionMismatch(f"PDControl gains must be ({f.nv},)")
            state._check(control_.q_des, f.nq, "q_des")
            state._check(control_.torques, f.nv, "torques")
            ctl.kind, ctl.kp, ctl.kd = _capi.CONTROL_PD, kp.data_ptr(), kd.data_ptr()
            ctl.q_des = control_.q_des.data_ptr() if control_.q_des is not None else None
            ctl.tau = control_.torques.data_ptr() if control_.torques is not None else None
            keep = (kp, kd)
        if store:
            raise NotImplementedError("store=True with a device-side controller: call simulate_ once per step")
        _raise(L.rbd_simulate_controlled(state.ws.handle, state.batch, _ptr(state.q), _ptr(state.v), ctypes.byref(ctl), _ptr(externalwrenches),
                                         ctypes.c_double(dt), nsteps, ctypes.byref(opts)), "rbd_simulate_controlled")
        del keep
    elif control_ is None and not store:
        _raise(L.rbd_simulate(state.ws.handle, state.batch, _ptr(state.q), _ptr(state.v), _ptr(torques), _ptr(externalwrenches),
                              ctypes.c_double(dt), nsteps, ctypes.byref(opts)), "rbd_simulate")
    elif control_ is None:
        for _ in range(nsteps):
            _raise(L.rbd_simulate(state.ws.handle, state.batch, _ptr(state.q), _ptr(state.v), _ptr(torques), _ptr(externalwrenches),
                                  ctypes.c_double(dt), 1, ctypes.byref(opts)), "rbd_simulate")
            qs.append(state.q.clone())
            vs.append(state.v.clone())
    else:
        tau = torch.zeros_like(state.v) if torques is None else torques
        vd = torch.zeros_like(state.v)
        lam = torch.zeros((state.batch, max(f.nc, 1)), dtype=state.dtype, device=state.device)
        for k in range(nsteps):
            for stage in range(5):
                _raise(L.rbd_mk_stage(state.ws.handle, state.batch, stage, ctypes.c_double(dt), _ptr(state.q), _ptr(state.v),
                                      _ptr(vd if stage > 0 else None), ctypes.byref(opts)), "rbd_mk_stage")
                if stage < 4:
                    control_(tau, ts[k] + RK4_C[stage] * dt, state)
                    _raise(L.rbd_dynamics(state.ws.handle, state.batch, _ptr(state.q), _ptr(state.v), _ptr(tau), _ptr(externalwrenches),
                                          _ptr(vd), None, _ptr(lam if f.nc else None), ctypes.byref(opts)), "rbd_dynamics")
            if store:
                qs.append(state.q.clone())
                vs.append(state.v.clone())
    ts = np.array(ts)
    return (ts, qs, vs) if store else ts


def dynamics_ode_(xd: torch.Tensor, result: DynamicsResult, state: MechanismState, x: torch.Tensor, torques: Optional[torch.Tensor] = None,
                  externalwrenches: Optional[torch.Tensor] = None, stabilization_gains="default"):
    """`dynamics!(ẋ, result, state, x, torques, externalwrenches)` (src/mechanism_algorithms.jl:880-889), the form used with
    off-the-shelf ODE solvers: x = [q; v; s] per state, (B, nq + nv + ns) with s the additional (contact) state — empty without contact
    points; `copyto!(state, x)`, `dynamics!`, `copyto!(ẋ, result)` with ẋ = [q̇; v̇; ṡ] (src/mechanism_state.jl:419-426,
    src/dynamics_result.jl:89-95)."""
    f = state.flat
    ns = getattr(f, "ns", 0)
    if state.layout != "aos":
        raise ValueError("dynamics_ode_ expects the AOS layout (one state vector per row)")
    n = f.nq + f.nv + ns
    if tuple(x.shape) != (state.batch, n) or tuple(xd.shape) != (state.batch, n):
        raise DimensionMismatch(f"x / ẋ must be ({state.batch}, {n}) = [q; v{'; s' if ns else ''}]")
    state.q.copy_(x[:, :f.nq])
    state.v.copy_(x[:, f.nq:f.nq + f.nv])
    if ns:
        state.s.copy_(x[:, f.nq + f.nv:])
    dynamics_(result, state, torques, externalwrenches, stabilization_gains=stabilization_gains)
    xd[:, :f.nq].copy_(result.qd)
    xd[:, f.nq:f.nq + f.nv].copy_(result.vd)
    if ns:
        xd[:, f.nq + f.nv:].copy_(result.sd)
    return xd


def _kin(state: MechanismState, A=None, com=None, energy=None):
    state.ws.use_current_stream()
    opts = state._opts()
    _raise(_capi.lib().rbd_kinematics(state.ws.handle, state.batch, _ptr(state.q), _ptr(state.v), _ptr(A), _ptr(com), _ptr(energy),
                                      ctypes.byref(opts)), "rbd_kinematics")


def momentum_matrix_(out: torch.Tensor, state: MechanismState):
    """`momentum_matrix!(out, state)` in the root frame (src/mechanism_algorithms.jl:313-327): out is (B, 6*nv), each row a 6×nv
    column-major matrix whose column i is crb_inertia(body(i))·S_i, (angular; linear)."""
    state._check(out, 6 * state.flat.nv, "momentum matrix")
    _kin(state, A=out)
    return out


def center_of_mass(state: MechanismState) -> torch.Tensor:
    """`center_of_mass(state)` in the root frame (src/mechanism_algorithms.jl:28-50): (B, 3)."""
    out = state._zeros(3)
    _kin(state, com=out)
    return out


def kinetic_energy(state: MechanismState) -> torch.Tensor:
    """`kinetic_energy(state)` (src/mechanism_state.jl:886-888, :989-994): (B,)."""
    e = state._zeros(2)
    _kin(state, energy=e)
    return e[:, 0] if state.layout == "aos" else e[0]


def gravitational_potential_energy(state: MechanismState) -> torch.Tensor:
    """`gravitational_potential_energy(state)` (src/mechanism_state.jl:897-903, :996-1000): (B,)."""
    e = state._zeros(2)
    _kin(state, energy=e)
    return e[:, 1] if state.layout == "aos" else e[1]


def jit_source(flat, dtype=torch.float32, family="mass_matrix"):
    """The source `rbd_jit_source` generates for a mechanism's run-time specialised kernels (csrc/rbd_jit.hip, rbd_spec.hpp); None when the
    mechanism is outside the one-lane-per-state kernels' scope.  family: "mass_matrix" (+ Cholesky), "dynamics", "inverse_dynamics" — one program each.  Host only."""
    L = _capi.lib()
    h = ctypes.c_void_p()
    _raise(L.rbd_model_create(ctypes.cast(ctypes.byref(flat.c_struct()), ctypes.c_void_p), ctypes.byref(h)), "rbd_model_create")
    try:
        dt = _capi.F64 if dtype == torch.float64 else _capi.F32
        fam = {"mass_matrix": 0, "dynamics": 1, "inverse_dynamics": 2, "loops": 3, "dynamics_tracks": 4, "inverse_dynamics_tracks": 5, "dynamics_tracks_pairs": 6,
               "inverse_dynamics_tracks_pairs": 7, "banked": 8, "dynamics_tracks_sim": 9, "dynamics_tracks_pairs_sim": 10, "kinematics": 11}[family]
        n = L.rbd_jit_source(h, dt, fam, None, 0)
        if n < 0:
            return None
        buf = ctypes.create_string_buffer(n + 1)
        L.rbd_jit_source(h, dt, fam, buf, n + 1)
        return buf.value.decode()
    finally:
        L.rbd_model_destroy(h)


def jit_precompile(flat, dtype=torch.float32):
    """Compiles a mechanism's specialised kernels into the on-disk cache ahead of time (`rbd_jit_precompile`; no device needed).  Returns
    (ok, compiler log); ok is None when the mechanism has no specialised kernels or hiprtc is not available."""
    L = _capi.lib()
    h = ctypes.c_void_p()
    _raise(L.rbd_model_create(ctypes.cast(ctypes.byref(flat.c_struct()), ctypes.c_void_p), ctypes.byref(h)), "rbd_model_create")
    try:
        log = ctypes.create_string_buffer(1 << 16)
        st = L.rbd_jit_precompile(h, _capi.F64 if dtype == torch.float64 else _capi.F32, log, len(log))
        return (None if st == 3 else st == 0), log.value.decode(errors="replace")
    finally:
        L.rbd_model_destroy(h)


def jit_status(flat, dtype=torch.float32, family: int = 0) -> int:
    """`rbd_jit_status`: 1 = the program's code object is ready, 0 = being compiled on a background thread (started by this call if nobody had),
    -1 = no such program for this mechanism / no hiprtc / the compilation failed.  Never waits; no device needed."""
    L = _capi.lib()
    h = ctypes.c_void_p()
    _raise(L.rbd_model_create(ctypes.cast(ctypes.byref(flat.c_struct()), ctypes.c_void_p), ctypes.byref(h)), "rbd_model_create")
    try:
        return int(L.rbd_jit_status(h, _capi.F64 if dtype == torch.float64 else _capi.F32, int(family)))
    finally:
        L.rbd_model_destroy(h)


def chain_plan(flat):
    """The chain schedule under the track / walk plans of a mechanism (host-side introspection of `rbd_model_chain_plan`): dict with `tracks`,
    `steps`, `lds_fields` and `table` (steps × tracks array of body indices, -1 = idle); None when the mechanism is outside
    that mapping's scope."""
    import numpy as np
    L = _capi.lib()
    h = ctypes.c_void_p()
    _raise(L.rbd_model_create(ctypes.cast(ctypes.byref(flat.c_struct()), ctypes.c_void_p), ctypes.byref(h)), "rbd_model_create")
    try:
        g, ns, nf = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
        st = L.rbd_model_chain_plan(h, ctypes.byref(g), ctypes.byref(ns), ctypes.byref(nf), None, 0)
        if st == 3:  # RBD_ERR_UNSUPPORTED
            return None
        _raise(st, "rbd_model_chain_plan")
        tab = np.zeros(ns.value * g.value, np.int32)
        _raise(L.rbd_model_chain_plan(h, None, None, None, tab.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), tab.size), "rbd_model_chain_plan")
        return {"tracks": g.value, "steps": ns.value, "lds_fields": nf.value, "table": tab.reshape(ns.value, g.value)}
    finally:
        L.rbd_model_destroy(h)


def reroot_plan(flat):
    """The walk kernels' plan for the tree re-rooted at its centre (`rbd_model_reroot_plan`, host-only; csrc/rbd_reroot.hpp): dict with `tracks`,
    `steps`, `root` / `floating_body` (body indices), the packed records `ri` / `rr`, the parking words `wk` and the chain table; None when the
    mechanism is not re-rooted (no floating base, already as shallow as it gets, chain too long or with joints that cannot be reversed)."""
    import numpy as np
    L = _capi.lib()
    h = ctypes.c_void_p()
    _raise(L.rbd_model_create(ctypes.cast(ctypes.byref(flat.c_struct()), ctypes.c_void_p), ctypes.byref(h)), "rbd_model_create")
    try:
        dims = np.zeros(12, np.int32)
        pi = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
        pd = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
        st = L.rbd_model_reroot_plan(h, pi(dims), None, 0, None, 0, None, 0, None, None, None)
        if st == 3:  # RBD_ERR_UNSUPPORTED
            return None
        _raise(st, "rbd_model_reroot_plan")
        n, nch = int(dims[0]) * int(dims[1]), int(dims[7])
        ri, rr, wk = np.zeros(n * 4 + int(dims[1]), np.int32), np.zeros(n * 24, np.float64), np.zeros(n, np.int32)
        ci, cr, fxp = np.zeros(max(4 * nch, 1), np.int32), np.zeros(max(15 * nch, 1), np.float64), np.zeros(12, np.float64)
        _raise(L.rbd_model_reroot_plan(h, None, pi(ri), ri.size, pd(rr), rr.size, pi(wk), wk.size, pi(ci), pd(cr), pd(fxp)), "rbd_model_reroot_plan")
        return {"tracks": int(dims[0]), "steps": int(dims[1]), "root": int(dims[10]), "floating_body": int(dims[11]), "chain": nch, "dims": dims,
                "ri": ri, "rr": rr, "wk": wk, "chain_i": ci, "chain_r": cr, "fxp": fxp}
    finally:
        L.rbd_model_destroy(h)


def track_plan(flat):
    """The track-mapping plan of a mechanism (`rbd_model_track_plan`, host-only): dict with `tracks`, `steps`, `mailboxes` (A/C, B),
    `floating`, `general`, `table` (steps × tracks body indices, -1 = idle) and the packed records `ri` / `rr` the kernel reads;
    None when the mechanism is outside the mapping's scope."""
    import numpy as np
    L = _capi.lib()
    h = ctypes.c_void_p()
    _raise(L.rbd_model_create(ctypes.cast(ctypes.byref(flat.c_struct()), ctypes.c_void_p), ctypes.byref(h)), "rbd_model_create")
    try:
        dims = np.zeros(6, np.int32)
        pi = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
        st = L.rbd_model_track_plan(h, pi(dims), None, 0, None, 0, None, 0)
        if st == 3:  # RBD_ERR_UNSUPPORTED
            return None
        _raise(st, "rbd_model_track_plan")
        n = int(dims[0]) * int(dims[1])
        tab, ri, rr = np.zeros(n, np.int32), np.zeros(n * 4 + int(dims[1]), np.int32), np.zeros(n * 24, np.float64)  # ri: packed records + per-step flags
        _raise(L.rbd_model_track_plan(h, None, pi(tab), tab.size, pi(ri), ri.size, rr.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), rr.size), "rbd_model_track_plan")
        return {"tracks": int(dims[0]), "steps": int(dims[1]), "mailboxes": (int(dims[2]), int(dims[3])), "floating": bool(dims[4]), "general": bool(dims[5]),
                "dims": dims, "table": tab.reshape(int(dims[1]), int(dims[0])), "ri": ri, "rr": rr}
    finally:
        L.rbd_model_destroy(h)


def geometric_jacobian_(out: torch.Tensor, state: MechanismState, base: int, body: int):
    """`geometric_jacobian!(out, state, path(mechanism, base, body))` in the root frame (src/mechanism_algorithms.jl:80-99):
    out is (B, 6*nv), each row a 6×nv column-major matrix, (angular; linear); `base` / `body` are body indices of the flat
    model (-1 = the root body).  J·v is `relative_twist(state, body, base)`."""
    state._check(out, 6 * state.flat.nv, "jacobian")
    state.ws.use_current_stream()
    opts = state._opts()
    _raise(_capi.lib().rbd_geometric_jacobian(state.ws.handle, state.batch, _ptr(state.q), int(base), int(body), _ptr(out), ctypes.byref(opts)),
           "rbd_geometric_jacobian")
    return out


def _momentum(state: MechanismState) -> torch.Tensor:
    out = state._zeros(12)
    state.ws.use_current_stream()
    opts = state._opts()
    _raise(_capi.lib().rbd_momentum(state.ws.handle, state.batch, _ptr(state.q), _ptr(state.v), _ptr(out), ctypes.byref(opts)), "rbd_momentum")
    return out if state.layout == "aos" else out.t()


def momentum(state: MechanismState) -> torch.Tensor:
    """`momentum(state)` in the root frame (src/mechanism_state.jl:975-980): (B, 6) = (angular; linear)."""
    return _momentum(state)[:, :6]


def momentum_rate_bias(state: MechanismState) -> torch.Tensor:
    """`momentum_rate_bias(state)` (src/mechanism_state.jl:982-987): (B, 6) wrench (torque; force); d/dt momentum = A v̇ + this."""
    return _momentum(state)[:, 6:]


def bank_plan(flat):
    """The two-bodies-per-lane plan of a mechanism (`rbd_model_bank_plan`): dict(lanes, L0, bodies=(n0, n1), aba) or None when the split
    would not pack more states into a wavefront.  Host-only."""
    L = _capi.lib()
    h = ctypes.c_void_p()
    _raise(L.rbd_model_create(ctypes.cast(ctypes.byref(flat.c_struct()), ctypes.c_void_p), ctypes.byref(h)), "rbd_model_create")
    try:
        v = [ctypes.c_int32() for _ in range(5)]
        st = L.rbd_model_bank_plan(h, *[ctypes.byref(x) for x in v])
        if st == 3:
            return None
        _raise(st, "rbd_model_bank_plan")
        return {"lanes": v[0].value, "L0": v[1].value, "bodies": (v[2].value, v[3].value), "aba": bool(v[4].value)}
    finally:
        L.rbd_model_destroy(h)
