"""which entries of the emitted M differ from the oracle's (debugging aid for chol_spec's emission)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle, rbd_amd as rbd
model = rbd.load_flat_model(os.path.join(ROOT, "tests", "golden", "models", "atlas_floating.json"))
B = 65536
rng = np.random.default_rng(0)
q = rbd.rand_configuration(model, B, rng); v = rbd.rand_velocity(model, B, rng); tau = rng.random((B, model.nv))
st = rbd.MechanismState(model, B, dtype=torch.float32, device="cuda:0")
x = torch.zeros(B, model.nv, dtype=torch.float32, device="cuda:0")
Mo = torch.zeros(B, model.nv * model.nv, dtype=torch.float32, device="cuda:0")
rbd.set_configuration_(st, q); rbd.set_velocity_(st, v)
rbd.mass_matrix_solve_(x, st, torch.as_tensor(tau, dtype=torch.float32).cuda(), M_out=Mo)
torch.cuda.synchronize()
print(rbd.last_kernel(st))
M = Mo.cpu().numpy().astype(np.float64).reshape(B, model.nv, model.nv)
for s in (0, 5, 17, B - 1):
    ref = oracle.mass_matrix(model, q[s:s + 1])[0]
    ref = np.tril(ref) + np.tril(ref, -1).T
    d = np.abs(M[s] - ref) > 1e-4 * np.abs(ref).max()
    print("state", s, "bad entries", int(d.sum()), "transposed-match", np.abs(M[s].T - ref).max() < 1e-3)
    if d.any():
        ij = np.argwhere(d)
        print(ij[:40].tolist())
        i, j = ij[0]
        print("got", M[s][i, j], "ref", ref[i, j], "got^T", M[s][j, i])
