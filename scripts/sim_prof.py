"""`simulate` wall time per RK4 step; run under `rocprofv3 --kernel-trace --stats` for the per-kernel split.  usage: sim_prof.py [B] [f64|f32]"""
import os, sys, time
os.environ.setdefault("RBD_JIT_ASYNC", "0")  # wait for the kernels compiled per mechanism instead of starting on the interpreting ones
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import torch
import rbd_amd as rbd
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dt = torch.float32 if len(sys.argv) > 2 and sys.argv[2] == "f32" else torch.float64
model = rbd.load_flat_model(os.path.join(ROOT, "tests", "golden", "models", "atlas_floating.json"))
state = rbd.MechanismState(model, B, dtype=dt); rbd.rand_(state, seed=1)
rbd.simulate_(state, 0.0095, dt=1e-3); torch.cuda.synchronize()
n = 50
t0 = time.perf_counter(); rbd.simulate_(state, (n - 0.5) * 1e-3, dt=1e-3); torch.cuda.synchronize(); t1 = time.perf_counter()
print("B", B, dt, "us per step", round((t1 - t0) / n * 1e6, 1), "kernel", rbd.last_kernel(state))
