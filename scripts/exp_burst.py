"""Per-launch HIP-event times of the headline kernel in a burst after idle: why 5 warm-up + 20 timed steps average 21 us when 2000 average 19.4."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import rbd_amd as rbd
model = rbd.load_flat_model(os.path.join(ROOT, "tests", "golden", "models", "atlas_floating.json"))
B = 4096
rng = np.random.default_rng(0)
state = rbd.MechanismState(model, B); result = rbd.DynamicsResult(model, B)
rbd.set_configuration_(state, rbd.rand_configuration(model, B, rng)); rbd.set_velocity_(state, rbd.rand_velocity(model, B, rng))
tau = torch.rand(B, model.nv, dtype=torch.float64, device="cuda")
f = lambda: rbd.dynamics_(result, state, tau)
for trial in range(3):
    torch.cuda.synchronize(); time.sleep(0.5 if trial else 0.0)
    n = 40
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        f(); ev[i + 1].record()
    torch.cuda.synchronize()
    ts = [ev[i].elapsed_time(ev[i + 1]) * 1000 for i in range(n)]
    print("burst", trial, " ".join("%.1f" % t for t in ts), flush=True)
# the same burst without events in between: 5 + 20 with one pair of events
for trial in range(3):
    torch.cuda.synchronize(); time.sleep(0.5)
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    print("5 + 20:", round(e0.elapsed_time(e1) * 1000 / 20, 2), "us per step", flush=True)
for trial in range(2):
    torch.cuda.synchronize(); time.sleep(0.5)
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    print("5 + 20 without the sync after warm-up:", round(e0.elapsed_time(e1) * 1000 / 20, 2), "us per step", flush=True)
