"""Phase timeline of aba_walk_kernel from in-kernel clock64() marks (thread 0 of block 0 = the longest track).  Needs the profiling build:
   OUT=librbd_hip_prof.so rigidbodydynamics.jl_amd/csrc/build.sh -DRBD_PROFILE_PHASES ; RBD_LIB=<that .so> python scripts/walk_phases.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import rbd_amd as rbd
from rigidbodydynamics_jl_amd import _capi
model = rbd.load_flat_model(os.path.join(ROOT, "tests", "golden", "models", "atlas_floating.json"))
for dt, tdt, B, pair in (("f64", torch.float64, 4096, False), ("f64", torch.float64, 65536, False), ("f32", torch.float32, 4096, False),
                         ("f32", torch.float32, 65536, False), ("f32", torch.float32, 65536, True)):
    os.environ["RBD_TUNE"] = "walk_pair_min_batch=" + ("1" if pair else str(1 << 40))
    rng = np.random.default_rng(1)
    state = rbd.MechanismState(model, B, dtype=tdt); result = rbd.DynamicsResult(model, B, dtype=tdt)
    rbd.set_configuration_(state, rbd.rand_configuration(model, B, rng)); rbd.set_velocity_(state, rbd.rand_velocity(model, B, rng))
    tau = torch.rand(B, model.nv, dtype=tdt, device="cuda")
    for _ in range(5): rbd.dynamics_(result, state, tau, algorithm="aba_walk")
    torch.cuda.synchronize()
    out = (ctypes.c_longlong * 32)()
    assert _capi.lib().rbd_debug_walk_phase_clock(out) == 0
    for g in range(4):  # one line per wavefront (= track) of block 0; all clocks relative to wave 0's start
        t = list(out)[8 * g:8 * g + 8]
        print(dt, "two states per lane" if pair else "", "B", B, "track", g, "start", t[0] - out[0], "cycles: staging in", t[1] - t[0], "pass A", t[2] - t[1], "pass B", t[3] - t[2],
              "pass C", t[4] - t[3], "out", t[5] - t[4], "total", t[5] - t[0], flush=True)
