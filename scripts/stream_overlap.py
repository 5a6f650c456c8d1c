"""Experiment: independent batches issued round-robin on N HIP streams (one workspace per stream) — how much of the
per-launch start-up latency overlaps with the previous launch's tail.  usage: stream_overlap.py [B] [nstreams]"""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import rbd_amd as rbd
from rigidbodydynamics_jl_amd import _capi
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 2
model = rbd.load_flat_model(os.path.join(ROOT, "tests", "golden", "models", "atlas_floating.json"))
L = _capi.lib()
sets = []
for i in range(NS):
    rng = np.random.default_rng(i)
    st = rbd.MechanismState(model, B); res = rbd.DynamicsResult(model, B)
    rbd.set_configuration_(st, rbd.rand_configuration(model, B, rng)); rbd.set_velocity_(st, rbd.rand_velocity(model, B, rng))
    tau = torch.rand(B, model.nv, dtype=torch.float64, device="cuda")
    s = torch.cuda.Stream()
    L.rbd_workspace_set_stream(st.ws.handle, ctypes.c_void_p(s.cuda_stream))
    opts = st._opts()
    args = (st.ws.handle, B, ctypes.c_void_p(st.q.data_ptr()), ctypes.c_void_p(st.v.data_ptr()), ctypes.c_void_p(tau.data_ptr()), ctypes.c_void_p(0),
            ctypes.c_void_p(res.vd.data_ptr()), ctypes.c_void_p(res.qd.data_ptr()), ctypes.c_void_p(0), ctypes.byref(opts))
    sets.append((st, res, tau, s, opts, args))
torch.cuda.synchronize()
K = 4000
for w in range(200):
    L.rbd_dynamics(*sets[w % NS][5])
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(K):
    L.rbd_dynamics(*sets[k % NS][5])
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(json.dumps({"batch": B, "streams": NS, "us_per_step": round(dt / K * 1e6, 2), "Mevals_per_s": round(B * K / dt / 1e6, 1)}))
