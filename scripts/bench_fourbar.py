"""BASELINE configs[4]: four-bar linkage (loop joint), batch 4096 fp64 — µs per dynamics! launch (RNEA + CRBA + loop solve), graph-replayed."""
import json, os, sys
os.environ.setdefault("RBD_JIT_ASYNC", "0")  # wait for the kernels compiled per mechanism instead of starting on the interpreting ones
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import rbd_amd as rbd
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
model = rbd.flatten(rbd.four_bar_linkage())
rng = np.random.default_rng(3)
q = np.tile(rbd.FOUR_BAR_INITIAL_Q, (B, 1)); q[:, 0] += rng.uniform(-0.05, 0.05, B)
v = np.tile(rbd.FOUR_BAR_INITIAL_V, (B, 1))
state = rbd.MechanismState(model, B); res = rbd.DynamicsResult(model, B)
rbd.set_configuration_(state, q); rbd.set_velocity_(state, v)
tau = torch.rand(B, model.nv, dtype=torch.float64, device="cuda")
f = lambda: rbd.dynamics_(res, state, tau)
for _ in range(3): f()
torch.cuda.synchronize()
reps = 100
g = torch.cuda.CUDAGraph(); cap = torch.cuda.Stream()
with torch.cuda.stream(cap):
    f()
    with torch.cuda.graph(g, stream=cap):
        for _ in range(reps): f()
torch.cuda.synchronize(); g.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / reps * 1e3
print(json.dumps({"model": "four_bar", "batch": B, "dtype": "f64", "us_per_launch": round(us, 2), "Mevals_per_s": round(B / us, 2)}))
