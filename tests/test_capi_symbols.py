"""CPU checks of the boundary: librbd_hip.so loads, exports every symbol include/rbd_hip.h declares, validates
models, and refuses to run without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest


def test_header_symbols_exported(rbd):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "rbd_hip.h")).read()
    declared = set(re.findall(r"\b(rbd_[a-z_]+)\s*\(", header))
    assert declared == set(rbd._capi.SYMBOLS)
    L = rbd._capi.lib()
    for s in declared:
        assert hasattr(L, s), s
    assert L.rbd_version() >= 100
    assert L.rbd_status_string(2) == b"dimension mismatch"


def test_model_create_validates(rbd, models):
    from rigidbodydynamics_jl_amd.state import _Model
    m = _Model(models["atlas_floating"])
    dims = [ctypes.c_int32() for _ in range(4)]
    assert rbd._capi.lib().rbd_model_dims(m.handle, *[ctypes.byref(d) for d in dims]) == 0
    assert [d.value for d in dims] == [31, 37, 36, 0]
    m4 = _Model(models["four_bar"])
    assert rbd._capi.lib().rbd_model_dims(m4.handle, *[ctypes.byref(d) for d in dims]) == 0
    assert [d.value for d in dims] == [3, 3, 3, 5]
    # parents must come first
    bad = rbd.flatten(rbd.double_pendulum())
    bad.parent = np.array([1, -1], dtype=np.int32)
    bad._c = None
    with pytest.raises(ValueError):
        _Model(bad)
    # q ranges must be consistent with the joint types
    bad = rbd.flatten(rbd.double_pendulum())
    bad.q_offset = np.array([0, 2], dtype=np.int32)
    bad._c = None
    with pytest.raises(rbd.DimensionMismatch):
        _Model(bad)


def test_no_cpu_fallback(rbd, models):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        rbd.MechanismState(models["double_pendulum"], 4)
    with pytest.raises(RuntimeError):
        rbd.MechanismState(models["double_pendulum"], 4, device="cpu")
    # straight through the C ABI: workspace creation reports RBD_ERR_NO_DEVICE
    from rigidbodydynamics_jl_amd.state import _Model
    m = _Model(models["double_pendulum"])
    h = ctypes.c_void_p()
    st = rbd._capi.lib().rbd_workspace_create(m.handle, 4, 0, 0, None, ctypes.byref(h))
    assert st == 4


def test_julia_shim_ccalls_match_the_header():
    """julia/RigidBodyDynamicsGPU.jl cannot be executed here (no Julia): its ccall signatures are checked against include/rbd_hip.h."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "scripts", "check_julia_ccalls.py")], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
