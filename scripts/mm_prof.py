"""(developer) kernel-level view of mass_matrix! + Cholesky solve at a large batch: run under rocprofv3 --kernel-trace --stats; argv: f32|f64 [batch]"""
import os, sys
os.environ.setdefault("RBD_JIT_ASYNC", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch, rbd_amd as rbd
tdt = torch.float64 if len(sys.argv) > 1 and sys.argv[1] == "f64" else torch.float32
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
model = rbd.load_flat_model(os.path.join(ROOT, "tests/golden/models/atlas_floating.json"))
rng = np.random.default_rng(1)
state = rbd.MechanismState(model, B, dtype=tdt); result = rbd.DynamicsResult(model, B, dtype=tdt)
rbd.set_configuration_(state, rbd.rand_configuration(model, B, rng)); rbd.set_velocity_(state, rbd.rand_velocity(model, B, rng))
tau = torch.rand(B, model.nv, dtype=tdt, device="cuda"); out = torch.zeros_like(tau)
for _ in range(20): rbd.mass_matrix_solve_(out, state, tau, result.massmatrix)
torch.cuda.synchronize()
print(rbd.last_kernel(state))
