"""ctypes binding of csrc/librbd_hip.so (include/rbd_hip.h).  The product path: if the shared library is
missing or no HIP device is visible, everything here raises — there is NO CPU fallback."""
from __future__ import annotations

import atexit
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RBD_LIB") or os.path.join(_HERE, "csrc", "librbd_hip.so")  # RBD_LIB: A/B kernel variants (experiments only)

RBD_OK = 0
HEADER_VERSION = 600  # RBD_HIP_H_VERSION of the include/rbd_hip.h this binding was written against
F64, F32 = 0, 1
LAYOUT_SOA, LAYOUT_AOS = 0, 1
MEM_DEVICE, MEM_HOST = 0, 1
ALGO_ABA, ALGO_CRBA_CHOLESKY, ALGO_ABA_LANES, ALGO_ABA_CHAINS, ALGO_ABA_BANKS, ALGO_ABA_TRACKS, ALGO_ABA_WALK, ALGO_ABA_PIPE, ALGO_ABA_COMPILED = 0, 1, 2, 3, 4, 5, 6, 7, 8

# every symbol include/rbd_hip.h declares (checked by tests/test_capi_symbols.py)
SYMBOLS = (
    "rbd_model_create", "rbd_model_destroy", "rbd_model_dims", "rbd_workspace_create", "rbd_workspace_destroy",
    "rbd_workspace_set_stream", "rbd_sync", "rbd_dynamics", "rbd_inverse_dynamics", "rbd_dynamics_bias", "rbd_mass_matrix",
    "rbd_mass_matrix_solve", "rbd_dynamics_result", "rbd_status_string", "rbd_last_hip_error",
    "rbd_workspace_enable_timing", "rbd_workspace_last_kernel_ms", "rbd_version", "rbd_simulate", "rbd_mk_stage", "rbd_cholesky_solve", "rbd_kinematics", "rbd_model_chain_plan", "rbd_workspace_last_kernel", "rbd_geometric_jacobian", "rbd_momentum", "rbd_model_bank_plan", "rbd_model_track_plan", "rbd_inverse_dynamics_bodies", "rbd_dynamics_bias_bodies",
    "rbd_model_reroot_plan", "rbd_model_contact_dims", "rbd_contact_dynamics", "rbd_dynamics_contact", "rbd_simulate_contact",
    "rbd_workspace_bind_result", "rbd_workspace_set_loop_gains", "rbd_jit_precompile", "rbd_jit_source", "rbd_jit_status", "rbd_jit_wait_idle", "rbd_simulate_controlled", "rbd_comm_unique_id", "rbd_comm_create", "rbd_comm_destroy", "rbd_comm_info", "rbd_gather", "rbd_gatherv", "rbd_mass_matrix_solve_packed", "rbd_comm_last_error", "rbd_jit_check_walk_object",
)


class Opts(ctypes.Structure):
    _fields_ = [("layout", ctypes.c_int32), ("memory", ctypes.c_int32), ("algorithm", ctypes.c_int32), ("stabilization", ctypes.c_int32)]


class Control(ctypes.Structure):  # rbd_control_t
    _fields_ = [("kind", ctypes.c_int32), ("per_stage", ctypes.c_int32), ("tau", ctypes.c_void_p), ("q_des", ctypes.c_void_p), ("kp", ctypes.c_void_p),
                ("kd", ctypes.c_void_p)]


CONTROL_CONSTANT, CONTROL_TABLE, CONTROL_PD = 0, 1, 2


class RBDError(RuntimeError):
    def __init__(self, status, where, detail=""):
        self.status = status
        super().__init__(f"{where}: status {status} ({detail})")


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(f"{LIB_PATH} not built — run `python -c 'import __graft_entry__ as g; g.build()'` "
                                    "(rigidbodydynamics.jl_amd has no CPU fallback)")
        L = ctypes.CDLL(LIB_PATH)
        if L.rbd_version() != HEADER_VERSION:  # struct layouts (rbd_flat_model_t) change with the header version
            raise RuntimeError(f"{LIB_PATH} reports rbd_version {L.rbd_version()}, this binding is for {HEADER_VERSION}: rebuild (csrc/build.sh)")
        vp, i32 = ctypes.c_void_p, ctypes.c_int32
        L.rbd_status_string.restype = ctypes.c_char_p
        L.rbd_status_string.argtypes = [ctypes.c_int]
        L.rbd_last_hip_error.restype = ctypes.c_char_p
        L.rbd_model_create.argtypes = [vp, ctypes.POINTER(vp)]
        L.rbd_model_destroy.argtypes = [vp]
        L.rbd_model_dims.argtypes = [vp] + [ctypes.POINTER(i32)] * 4
        L.rbd_workspace_create.argtypes = [vp, i32, i32, i32, vp, ctypes.POINTER(vp)]
        L.rbd_workspace_destroy.argtypes = [vp]
        L.rbd_workspace_set_stream.argtypes = [vp, vp]
        L.rbd_workspace_bind_result.argtypes = [vp, vp, vp]
        L.rbd_workspace_set_loop_gains.argtypes = [vp, ctypes.POINTER(ctypes.c_double)]
        L.rbd_sync.argtypes = [vp]
        L.rbd_dynamics.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp, ctypes.POINTER(Opts)]
        L.rbd_inverse_dynamics.argtypes = [vp, i32, vp, vp, vp, vp, vp, ctypes.POINTER(Opts)]
        L.rbd_dynamics_bias.argtypes = [vp, i32, vp, vp, vp, vp, ctypes.POINTER(Opts)]
        L.rbd_inverse_dynamics_bodies.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp, ctypes.POINTER(Opts)]
        L.rbd_dynamics_bias_bodies.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, ctypes.POINTER(Opts)]
        L.rbd_mass_matrix.argtypes = [vp, i32, vp, vp, ctypes.POINTER(Opts)]
        L.rbd_mass_matrix_solve.argtypes = [vp, i32, vp, vp, vp, vp, ctypes.POINTER(Opts)]
        L.rbd_mass_matrix_solve_packed.argtypes = [vp, i32, vp, vp, vp, vp, ctypes.POINTER(Opts)]
        L.rbd_dynamics_result.argtypes = [vp, i32, vp, vp, vp, vp, ctypes.POINTER(Opts)]
        L.rbd_cholesky_solve.argtypes = [vp, i32, vp, vp, vp, vp, ctypes.POINTER(Opts)]
        L.rbd_model_chain_plan.argtypes = [vp, ctypes.POINTER(i32), ctypes.POINTER(i32), ctypes.POINTER(i32), ctypes.POINTER(i32), i32]
        L.rbd_model_track_plan.argtypes = [vp, ctypes.POINTER(i32), ctypes.POINTER(i32), i32, ctypes.POINTER(i32), i32, ctypes.POINTER(ctypes.c_double), i32]
        L.rbd_comm_unique_id.argtypes = [vp]
        L.rbd_comm_create.argtypes = [vp, i32, i32, i32, ctypes.POINTER(vp)]
        L.rbd_comm_destroy.argtypes = [vp]
        L.rbd_comm_info.argtypes = [vp, ctypes.POINTER(i32), ctypes.POINTER(i32)]
        L.rbd_gather.argtypes = [vp, i32, vp, vp, ctypes.c_int64, i32, vp]
        L.rbd_gatherv.argtypes = [vp, i32, vp, vp, ctypes.POINTER(ctypes.c_int64), i32, vp]
        L.rbd_comm_last_error.restype = ctypes.c_char_p
        L.rbd_workspace_last_kernel.argtypes = [vp]
        L.rbd_workspace_last_kernel.restype = ctypes.c_char_p
        L.rbd_geometric_jacobian.argtypes = [vp, i32, vp, i32, i32, vp, ctypes.POINTER(Opts)]
        L.rbd_momentum.argtypes = [vp, i32, vp, vp, vp, ctypes.POINTER(Opts)]
        L.rbd_model_bank_plan.argtypes = [vp] + [ctypes.POINTER(i32)] * 5
        L.rbd_kinematics.argtypes = [vp, i32, vp, vp, vp, vp, vp, ctypes.POINTER(Opts)]
        L.rbd_simulate.argtypes = [vp, i32, vp, vp, vp, vp, ctypes.c_double, i32, ctypes.POINTER(Opts)]
        L.rbd_mk_stage.argtypes = [vp, i32, i32, ctypes.c_double, vp, vp, vp, ctypes.POINTER(Opts)]
        L.rbd_simulate_controlled.argtypes = [vp, i32, vp, vp, ctypes.POINTER(Control), vp, ctypes.c_double, i32, ctypes.POINTER(Opts)]
        L.rbd_model_reroot_plan.argtypes = [vp, ctypes.POINTER(i32), ctypes.POINTER(i32), i32, ctypes.POINTER(ctypes.c_double), i32, ctypes.POINTER(i32), i32,
                                            ctypes.POINTER(i32), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
        L.rbd_model_contact_dims.argtypes = [vp] + [ctypes.POINTER(i32)] * 3
        L.rbd_contact_dynamics.argtypes = [vp, i32, vp, vp, vp, vp, vp, ctypes.POINTER(Opts)]
        L.rbd_dynamics_contact.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ctypes.POINTER(Opts)]
        L.rbd_simulate_contact.argtypes = [vp, i32, vp, vp, vp, vp, vp, ctypes.c_double, i32, ctypes.POINTER(Opts)]
        L.rbd_workspace_enable_timing.argtypes = [vp, i32]
        L.rbd_jit_precompile.argtypes = [vp, i32, ctypes.c_char_p, ctypes.c_int64]
        L.rbd_jit_source.argtypes = [vp, i32, i32, ctypes.c_char_p, ctypes.c_int64]
        L.rbd_jit_source.restype = ctypes.c_int64
        L.rbd_jit_status.argtypes = [vp, i32, i32]
        L.rbd_jit_check_walk_object.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int64, ctypes.c_char_p, ctypes.c_int64]
        L.rbd_jit_wait_idle.argtypes = []
        L.rbd_jit_wait_idle.restype = None
        atexit.register(L.rbd_jit_wait_idle)  # a compilation still running in the background: wait for it before the compiler's own teardown
        L.rbd_workspace_last_kernel_ms.argtypes = [vp, ctypes.POINTER(ctypes.c_float)]
        _lib = L
    return _lib


def check(status: int, where: str):
    if status != RBD_OK:
        L = lib()
        detail = L.rbd_status_string(status).decode()
        hip = L.rbd_last_hip_error().decode()
        if hip and (status in (4, 5, 6) or (status == 3 and "differs from the interpreting kernel" in hip)):  # (3: a compiled program the first-use check dropped)
            detail += "; " + hip
        raise RBDError(status, where, detail)
