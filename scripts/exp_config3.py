"""Timing experiments on config 3 (mass_matrix! + tile Cholesky, fp32, 65 536 states): graph-replayed µs per call.
OP = solve (default: mass_matrix! + Cholesky solve, M emitted) | solve_nom (M_out = None) | mm (mass_matrix! alone)."""
import os, sys
os.environ.setdefault("RBD_JIT_ASYNC", "0")  # wait for the kernels compiled per mechanism instead of starting on the interpreting ones
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import rbd_amd as rbd
model = rbd.load_flat_model(os.path.join(ROOT, "tests", "golden", "models", "atlas_floating.json"))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
op = os.environ.get("OP", "solve")
tdt = torch.float64 if os.environ.get("DTYPE") == "f64" else torch.float32
rng = np.random.default_rng(2)
state = rbd.MechanismState(model, B, dtype=tdt); result = rbd.DynamicsResult(model, B, dtype=tdt)
rbd.set_configuration_(state, rbd.rand_configuration(model, B, rng))
tau = torch.rand(B, model.nv, dtype=tdt, device="cuda"); out = torch.zeros_like(tau)
alg = os.environ.get("ALG", "cholesky")  # "aba": x from one articulated-body pass (M, when asked for, from mass_matrix!)
if op == "solve": f = lambda: rbd.mass_matrix_solve_(out, state, tau, result.massmatrix, algorithm=alg)
elif op == "solve_nom": f = lambda: rbd.mass_matrix_solve_(out, state, tau, None, algorithm=alg)
else: f = lambda: rbd.mass_matrix_(result, state)
for _ in range(5): f()
torch.cuda.synchronize()
cap = torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(cap):
    f()
    with torch.cuda.graph(g, stream=cap):
        for _ in range(10): f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
g.replay(); torch.cuda.synchronize()
e0.record()
for _ in range(5): g.replay()
e1.record(); torch.cuda.synchronize()
print(os.environ.get("TAG", ""), op, alg, "B", B, "us per call", round(e0.elapsed_time(e1) * 1000 / 50, 2), rbd.last_kernel(state), flush=True)
