import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, time
import rbd_amd as rbd
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
model = rbd.load_flat_model(os.path.join(ROOT, "tests", "golden", "models", "atlas_floating.json"))
B = 4096
state = rbd.MechanismState(model, B); rbd.rand_(state, seed=1)
rbd.simulate_(state, 0.0095, dt=1e-3); torch.cuda.synchronize()
t0 = time.perf_counter(); rbd.simulate_(state, 0.1995, dt=1e-3); torch.cuda.synchronize(); t1 = time.perf_counter()
print("us per step", (t1 - t0) / 200 * 1e6)
