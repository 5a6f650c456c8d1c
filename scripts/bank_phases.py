#!/usr/bin/env python3
"""Phase timeline of aba_bank_kernel from in-kernel clock64() marks (wave 0 of block 0).  Needs the profiling build:
OUT=librbd_hip_prof.so csrc/build.sh -DRBD_PROFILE_PHASES ; RBD_LIB=.../librbd_hip_prof.so python scripts/bank_phases.py [B] [dtype]"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import rbd_amd as rbd
from rbd_amd import _capi, flatio
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dt = torch.float64 if (len(sys.argv) < 3 or sys.argv[2] == "f64") else torch.float32
model = flatio.load_flat_model(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "models", "atlas_floating.json"))
state = rbd.MechanismState(model, B, dtype=dt)
rbd.rand_(state, seed=1)
res = rbd.DynamicsResult(model, B, dtype=dt)
if len(sys.argv) > 3 and sys.argv[3] == "simulate":  # timeline of the last fused launch of a simulate run (stage 3 of the last step)
    rbd.simulate_(state, 0.0195, dt=1e-3)
else:
    for _ in range(20):
        rbd.dynamics_(res, state, algorithm="aba_banks")
torch.cuda.synchronize()
L = _capi.lib()
buf = (ctypes.c_longlong * 16)()
L.rbd_debug_bank_phase_clock.argtypes = [ctypes.c_void_p]
assert L.rbd_debug_bank_phase_clock(buf) == 0
t = np.array(buf[:12], dtype=np.int64)
names = ["fetch+setup0 (loads arrive)", "FK sweep bank 0", "setup1", "cross FK + park + FK sweep bank 1", "terms bank 1", "backward bank 1",
         "cross hand-off + unpark/park", "terms bank 0", "backward bank 0", "forward bank 0", "forward bank 1"]
d = np.diff(t)
print(f"B={B} {dt}: total {t[-1]-t[0]} clock64 ticks")
for n, x in zip(names, d):
    print(f"  {n:40s} {x:8d}  {100.0*x/(t[-1]-t[0]):5.1f}%")
