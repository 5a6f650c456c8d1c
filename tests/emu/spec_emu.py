"""TEST INFRASTRUCTURE ONLY — host emulation of the kernels compiled per mechanism (csrc/rbd_spec.hpp through rbd_jit_source): the generated program is compiled
as plain C++ against tests/emu/spec_shim/hip/hip_runtime.h and run one lane at a time (tests/emu/spec_emu_main.inc).  What it checks is the ARITHMETIC of the
straight-line code hiprtc will compile — the plan tables, the limbs walked in lockstep, the folded constants — against the oracle, on the CPU."""
import ctypes
import hashlib
import os
import subprocess
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def available():
    return os.path.exists(CLANG)


def build(source: str, which: str) -> ctypes.CDLL:
    """which: ABA | RNEA_F32 | RNEA_F64 — the entry point to keep (the program holds one kernel family)."""
    main = open(os.path.join(ROOT, "tests", "emu", "spec_emu_main.inc")).read()
    # __shared__ arrays inside extern "C" kernels become function statics; the kernels' own `extern __shared__` (mass matrix) is not emulated
    if which.startswith("MASS"):  # the program's own kernels use dynamic LDS, matrix cores and DPP: the harness calls crba_spec itself
        source = source[:source.index('extern "C" __global__')]
    text = "#define RBD_EMU_%s 1\n#define RBD_SPEC_EMU 1\n#include <hip/hip_runtime.h>\n" % which + source + "\n" + main
    key = hashlib.sha256((text + open(os.path.join(ROOT, "rigidbodydynamics.jl_amd", "csrc", "rbd_spec.hpp")).read()
                          + open(os.path.join(ROOT, "rigidbodydynamics.jl_amd", "csrc", "rbd_device.hpp")).read()).encode()).hexdigest()[:16]
    d = os.path.join(tempfile.gettempdir(), "rbd_spec_emu")
    os.makedirs(d, exist_ok=True)
    so = os.path.join(d, "emu_%s.so" % key)
    if not os.path.exists(so):
        src = os.path.join(d, "emu_%s.cpp" % key)
        open(src, "w").write(text)
        cmd = [CLANG, "-x", "c++", "-std=c++17", "-O1", "-fPIC", "-shared", "-ffp-contract=fast", "-march=native", "-DRBD_JIT_COMPILE", "-Wno-everything",
               "-I", os.path.join(ROOT, "tests", "emu", "spec_shim"), "-I", os.path.join(ROOT, "rigidbodydynamics.jl_amd", "csrc"), "-I", os.path.join(ROOT, "include"),
               src, "-o", so + ".tmp"]
        subprocess.check_call(cmd)
        os.replace(so + ".tmp", so)
    return ctypes.CDLL(so)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def aba_f32(lib, model, q, v, tau, fext=None, want_qdot=False):
    """q (B, nq), v (B, nv), tau (B, nv) -> v̇ (B, nv) [, q̇ (B, nq)] through aba_spec_f32 on the host."""
    B = q.shape[0]
    qs = np.ascontiguousarray(q.T, dtype=np.float32); vs = np.ascontiguousarray(v.T, dtype=np.float32); ts = np.ascontiguousarray(tau.T, dtype=np.float32)
    fs = np.ascontiguousarray(fext.T, dtype=np.float32) if fext is not None else None
    vd = np.full((model.nv, B), np.nan, dtype=np.float32)
    qd = np.full((model.nq, B), np.nan, dtype=np.float32) if want_qdot else None
    g = np.asarray(model.gravity, dtype=np.float64)
    lib.emu_aba_f32(ctypes.c_long(B), _p(qs), _p(vs), _p(ts), _p(fs), _p(vd), _p(qd), _p(g))
    return (vd.T.copy(), qd.T.copy()) if want_qdot else vd.T.copy()


def aba_f64(lib, model, q, v, tau, fext=None, want_qdot=False, stash=False):
    """... through aba_spec_f64 (round 6: the program in doubles of the mechanisms no walk kernel takes), or (stash) aba_spec_gst_f64."""
    B = q.shape[0]
    qs = np.ascontiguousarray(q.T, dtype=np.float64); vs = np.ascontiguousarray(v.T, dtype=np.float64); ts = np.ascontiguousarray(tau.T, dtype=np.float64)
    fs = np.ascontiguousarray(fext.T, dtype=np.float64) if fext is not None else None
    vd = np.full((model.nv, B), np.nan); qd = np.full((model.nq, B), np.nan) if want_qdot else None
    g = np.asarray(model.gravity, dtype=np.float64)
    lib.emu_aba_f64(ctypes.c_long(B), _p(qs), _p(vs), _p(ts), _p(fs), _p(vd), _p(qd), _p(g), ctypes.c_int(1 if stash else 0))
    return (vd.T.copy(), qd.T.copy()) if want_qdot else vd.T.copy()


def rnea(lib, model, q, v, vdot, fext=None, dtype=np.float32, want_bodies=False):
    B = q.shape[0]
    qs = np.ascontiguousarray(q.T, dtype=dtype); vs = np.ascontiguousarray(v.T, dtype=dtype)
    ws = np.ascontiguousarray(vdot.T, dtype=dtype) if vdot is not None else None
    fs = np.ascontiguousarray(fext.T, dtype=dtype) if fext is not None else None
    tau = np.full((model.nv, B), np.nan, dtype=dtype)
    acc = np.full((6 * model.n_bodies, B), np.nan, dtype=dtype) if want_bodies else None
    jw = np.full((6 * model.n_bodies, B), np.nan, dtype=dtype) if want_bodies else None
    f = lib.emu_rnea_f32 if dtype == np.float32 else lib.emu_rnea_f64
    f(ctypes.c_long(B), _p(qs), _p(vs), _p(ws), _p(fs), _p(tau), _p(acc), _p(jw))
    return (tau.T.copy(), acc.T.copy(), jw.T.copy()) if want_bodies else tau.T.copy()


def crba(lib, model, q, dtype=np.float32, permuted=False):
    """q (B, nq) -> M (B, nv, nv) as crba_spec leaves it (lower triangle; permuted: position (max, min) of (PERM[row], PERM[col])), NaN where nothing was written."""
    B, nv = q.shape[0], model.nv
    qs = np.ascontiguousarray(q.T, dtype=dtype)
    M = np.full((nv * nv, B), np.nan, dtype=dtype)
    f = lib.emu_crba_f32 if dtype == np.float32 else lib.emu_crba_f64
    f(ctypes.c_long(B), _p(qs), _p(M), ctypes.c_int(1 if permuted else 0))
    return M.T.reshape(B, nv, nv).transpose(0, 2, 1).copy()  # [b, row, col]


def kin(lib, model, q, v, dtype=np.float64, path=None):
    """The by-product kernels (kin_spec / jac_spec / mom_spec) on the host: A (B, 6, nv), com (B, 3), energies (B, 2), J (B, 6, nv) of path = (jplus, jminus) bit
    masks over the bodies in depth-first order, momentum + momentum_rate_bias (B, 12)."""
    B, nv = q.shape[0], model.nv
    qs = np.ascontiguousarray(q.T, dtype=dtype); vs = np.ascontiguousarray(v.T, dtype=dtype)
    A = np.full((6 * nv, B), np.nan, dtype=dtype); com = np.full((3, B), np.nan, dtype=dtype); en = np.full((2, B), np.nan, dtype=dtype)
    J = np.full((6 * nv, B), np.nan, dtype=dtype) if path is not None else None
    mom = np.full((12, B), np.nan, dtype=dtype)
    jp, jm = path if path is not None else (0, 0)
    lib.emu_kin(ctypes.c_long(B), _p(qs), _p(vs), _p(A), _p(com), _p(en), _p(J), ctypes.c_ulonglong(jp), ctypes.c_ulonglong(jm), _p(mom))
    unp = lambda X: X.T.reshape(B, nv, 6).transpose(0, 2, 1).copy()
    return unp(A), com.T.copy(), en.T.copy(), (unp(J) if J is not None else None), mom.T.copy()
