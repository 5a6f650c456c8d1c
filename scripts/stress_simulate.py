"""One-off stress: `simulate` (Munthe-Kaas RK4 on the device, fused stages) on random trees of every joint type against the numpy
restatement of src/ode_integrators.jl:233-299, with the lane mapping forced either way (RBD_TUNE=bank_min_batch=<n>)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import rbd_amd as rbd, oracle, simulate_np
from test_gpu_parity import canon_q
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(4242)
TYPES = ["Revolute", "Prismatic", "Fixed", "SinCosRevolute", "Planar", "QuaternionSpherical", "QuaternionFloating"]
worst = {"q": 0.0, "v": 0.0}
used = {}
for trial in range(N):
    n = int(rng.integers(1, 10))
    types = [str(rng.choice(TYPES, p=[0.3, 0.15, 0.1, 0.1, 0.15, 0.1, 0.1])) for _ in range(n)]
    model = rbd.flatten(rbd.rand_tree_mechanism(rng, types))
    if model.nv == 0:
        continue
    os.environ["RBD_TUNE"] = "bank_min_batch=" + ("1" if trial % 2 else "1000000")  # read when the workspace is created
    B, dt, T = 3, 2e-3, 0.0075
    r2 = np.random.default_rng(trial)
    q, v, tau = rbd.rand_configuration(model, B, r2), rbd.rand_velocity(model, B, r2), r2.random((B, model.nv))
    state = rbd.MechanismState(model, B)
    rbd.set_configuration_(state, q); rbd.set_velocity_(state, v)
    rbd.simulate_(state, T, dt=dt, torques=torch.as_tensor(tau).cuda())
    k = rbd._capi.lib().rbd_workspace_last_kernel(state.ws.handle).decode()
    used[k] = used.get(k, 0) + 1
    _, q_ref, v_ref = simulate_np.simulate(model, q, v, T, dt, tau)
    eq = np.abs(canon_q(model, state.q.cpu().numpy()) - canon_q(model, q_ref)).max() / max(1.0, np.abs(q_ref).max())
    ev = np.abs(state.v.cpu().numpy() - v_ref).max() / max(1.0, np.abs(v_ref).max())
    worst["q"], worst["v"] = max(worst["q"], eq), max(worst["v"], ev)
    assert eq < 1e-9 and ev < 1e-8, (trial, types, k, eq, ev)
print(f"{N} random trees simulated ok; worst relative errors {worst}; kernels used {used}")
