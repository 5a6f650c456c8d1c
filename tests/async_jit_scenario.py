"""Scenario of tests/test_state_kernels.py::test_first_call_does_not_wait_for_the_compiler, run in a process of its own (the library keeps the
programs it has compiled in a process-wide table, so inside the test process the kernels of Atlas would already be there): a mechanism whose compiled
kernels are NOT in the cache (RBD_JIT_ASYNC=1, empty RBD_JIT_CACHE).  Prints one line per step; the last line is `RESULT {json}`."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
T0 = time.time()
def log(*a): print("[%6.2f]" % (time.time() - T0), *a, flush=True)
import numpy as np, torch
import rbd_amd as rbd
cache = os.environ["RBD_JIT_CACHE"]
model = rbd.load_flat_model(os.path.join(ROOT, "tests", "golden", "models", "atlas_floating.json"))
out = {"async": os.environ.get("RBD_JIT_ASYNC"), "hiprtc": True}
# a small batch never starts a compilation it would not use
small = rbd.MechanismState(model, 64, dtype=torch.float32)
rs = rbd.DynamicsResult(model, 64, dtype=torch.float32)
rbd.rand_(small, 1)
rbd.dynamics_(rs, small)
tv = torch.zeros_like(small.v)
rbd.inverse_dynamics_(tv, small, rs.vd)
rbd.sync(small)
time.sleep(1.0)
out["small_kernel"] = rbd.last_kernel(small)
out["files_after_small_calls"] = sorted(os.listdir(cache))
log("small batch:", out["small_kernel"], out["files_after_small_calls"])
B = 65536
state = rbd.MechanismState(model, B, dtype=torch.float32)
result = rbd.DynamicsResult(model, B, dtype=torch.float32)
rbd.rand_(state, 3)
tau = torch.rand((B, model.nv), dtype=torch.float32, device="cuda")
torch.cuda.synchronize()
calls = []
vd_first = None
for k in range(150):
    t0 = time.time()
    try:
        rbd.dynamics_(result, state, tau)
    except rbd._capi.RBDError as e:
        out["error"] = str(e); break
    rbd.sync(state)
    calls.append((round(time.time() - t0, 4), rbd.last_kernel(state)))
    log("call", k, calls[-1])
    if k == 0:
        vd_first = result.vd.clone()
    if "aba_spec_f32" in calls[-1][1]:
        break
    time.sleep(0.5)
out["first_call_s"], out["first_kernel"] = calls[0]
out["slowest_call_s"] = max(c[0] for c in calls)
out["last_kernel"] = calls[-1][1]
out["seconds_until_compiled"] = round(time.time() - T0, 1)
# the two kernels agree (fp32, an ill-conditioned solve: compare through the residual of M v̇ = τ − c on a sample)
import oracle
n = 512
q, v, t = state.q[:n].double().cpu().numpy(), state.v[:n].double().cpu().numpy(), tau[:n].double().cpu().numpy()
M = oracle.mass_matrix(model, q); Ms = np.tril(M) + np.transpose(np.tril(M, -1), (0, 2, 1))
c = oracle.dynamics_bias(model, q, v, None)
def berr(vd):
    vd = vd[:n].double().cpu().numpy()
    r = np.einsum("bij,bj->bi", Ms, vd) - (t - c)
    return float((np.linalg.norm(r, axis=1) / (np.linalg.norm(Ms, axis=(1, 2)) * np.linalg.norm(vd, axis=1) + np.linalg.norm(t - c, axis=1))).max())
out["backward_err_first"], out["backward_err_last"] = berr(vd_first), berr(result.vd)
out["files_at_end"] = sorted(os.listdir(cache))
print("RESULT " + json.dumps(out), flush=True)
