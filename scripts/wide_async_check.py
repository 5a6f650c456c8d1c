"""(developer, GPU) a mechanism only the COMPILED one-lane-per-state kernels take (3-dof joints) with compilation in the background (RBD_JIT_ASYNC=1): the first calls
run on the lane-per-body kernels, later ones on the compiled kernels; every call is checked against the oracle.  Prints which kernel served from which call on."""
import os, sys, time
os.environ["RBD_JIT_ASYNC"] = "1"
os.environ["RBD_TUNE"] = "state_min_batch=1,spec_aba_min_batch=1,spec_rnea_min_batch=1"
os.makedirs("/tmp/jc_async", mode=0o700, exist_ok=True)
os.environ["RBD_JIT_CACHE"] = "/tmp/jc_async"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch, rbd_amd as rbd, oracle
m = rbd.flatten(rbd.rand_tree_mechanism(np.random.default_rng(11), ["QuaternionFloating", "QuaternionSpherical", "Planar", "Revolute", "Revolute", "Prismatic", "QuaternionSpherical", "Revolute", "SinCosRevolute"]))
B = 200
rng = np.random.default_rng(5)
q, v = rbd.rand_configuration(m, B, rng), rbd.rand_velocity(m, B, rng)
tau = rng.random((B, m.nv))
seen = {}
t0 = time.time()
for dt, tol in ((torch.float32, 2e-5), (torch.float64, 1e-10)):
    state = rbd.MechanismState(m, B, dtype=dt); res = rbd.DynamicsResult(m, B, dtype=dt)
    rbd.set_configuration_(state, q); rbd.set_velocity_(state, v)
    T = torch.as_tensor(tau, dtype=dt).cuda(); x = torch.zeros_like(T); out = torch.zeros_like(T)
    Mr = oracle.mass_matrix(m, q); Ms = np.tril(Mr) + np.transpose(np.tril(Mr, -1), (0, 2, 1))
    ref_id = oracle.inverse_dynamics(m, q, v, tau, None)
    for it in range(400):
        rbd.mass_matrix_solve_(x, state, T, None)
        k1 = rbd.last_kernel(state).split(" (")[0]
        xg = x.double().cpu().numpy()
        r = np.einsum("bij,bj->bi", Ms, xg) - tau
        eta = (np.linalg.norm(r, axis=1) / (np.linalg.norm(Ms, axis=(1, 2)) * np.linalg.norm(xg, axis=1) + np.linalg.norm(tau, axis=1))).max()
        assert eta < (2e-5 if dt == torch.float32 else 1e-12), (it, k1, eta)
        rbd.inverse_dynamics_(out, state, T)
        k2 = rbd.last_kernel(state).split(" (")[0]
        e = np.abs(out.double().cpu().numpy() - ref_id).max() / np.abs(ref_id).max()
        assert e < tol, (it, k2, e)
        rbd.dynamics_(res, state, T)
        k3 = rbd.last_kernel(state).split(" (")[0]
        for k in (k1, k2, k3):
            if (str(dt), k) not in seen:
                seen[(str(dt), k)] = (it, round(time.time() - t0, 1))
        if "spec" in k1 and "spec" in k2 and ("spec" in k3 or dt == torch.float64):
            break
        time.sleep(0.1)
print(seen)
