"""Profiling driver: fp64 dynamics! / inverse_dynamics! (with the per-body outputs) at 65 536 Atlas states through the walk kernels compiled for the mechanism and
through the interpreting ones, 40 launches each (scripts/gpu_walk_profile.sh runs it under rocprofv3)."""
import os, sys
os.environ.setdefault("RBD_JIT_ASYNC", "0")  # wait for the kernels compiled per mechanism instead of starting on the interpreting ones
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import rbd_amd as rbd
model = rbd.load_flat_model(os.path.join(ROOT, "tests", "golden", "models", "atlas_floating.json"))
B = 65536
rng = np.random.default_rng(3)
q, v = rbd.rand_configuration(model, B, rng), rbd.rand_velocity(model, B, rng)
tau = torch.rand(B, model.nv, dtype=torch.float64, device="cuda"); out = torch.zeros_like(tau)
jw = torch.zeros(B, 6 * model.n_bodies, dtype=torch.float64, device="cuda"); acc = torch.zeros_like(jw)
for jit in ("1", "0"):
    os.environ["RBD_JIT"] = jit
    state = rbd.MechanismState(model, B); result = rbd.DynamicsResult(model, B)
    rbd.set_configuration_(state, q); rbd.set_velocity_(state, v)
    for _ in range(40): rbd.dynamics_(result, state, tau)
    print(rbd.last_kernel(state))
    for _ in range(40): rbd.inverse_dynamics_(out, state, tau, mapping="walk", jointwrenchesout=jw, accelerations=acc)
    print(rbd.last_kernel(state))
    torch.cuda.synchronize()
