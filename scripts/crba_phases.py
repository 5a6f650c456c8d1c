#!/usr/bin/env python3
"""Phase timeline of crba_kernel from in-kernel clock64() marks (profiling build, see scripts/bank_phases.py)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rbd_amd as rbd
from rbd_amd import _capi, flatio
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
model = flatio.load_flat_model(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "models", "atlas_floating.json"))
state = rbd.MechanismState(model, B)
rbd.rand_(state, seed=1)
res = rbd.DynamicsResult(model, B)
for _ in range(20):
    rbd.mass_matrix_(res, state)
torch.cuda.synchronize()
L = _capi.lib()
buf = (ctypes.c_longlong * 16)()
L.rbd_debug_crba_phase_clock.argtypes = [ctypes.c_void_p]
assert L.rbd_debug_crba_phase_clock(buf) == 0
t = np.array(buf[:6], dtype=np.int64)
names = ["load + local transform", "FK sweep", "inertia + composite sweep", "records + own force columns", "support-chain walk + stores"]
print(f"crba_kernel B={B}: total {t[-1]-t[0]} ticks")
for n, x in zip(names, np.diff(t)):
    print(f"  {n:32s} {x:8d}  {100.0*x/(t[-1]-t[0]):5.1f}%")
