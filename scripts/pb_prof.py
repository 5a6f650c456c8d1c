"""(developer) kernel-level view of inverse_dynamics! with per-body outputs: run under rocprofv3 --kernel-trace --stats; argv[1] = aos | soa, argv[2] = f32 | f64"""
import os, sys
os.environ.setdefault("RBD_JIT_ASYNC", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch, rbd_amd as rbd
layout = sys.argv[1] if len(sys.argv) > 1 else "aos"
tdt = torch.float64 if len(sys.argv) > 2 and sys.argv[2] == "f64" else torch.float32
model = rbd.load_flat_model(os.path.join(ROOT, "tests/golden/models/atlas_floating.json"))
B = 65536
rng = np.random.default_rng(1)
state = rbd.MechanismState(model, B, dtype=tdt, layout=layout)
rbd.set_configuration_(state, rbd.rand_configuration(model, B, rng)); rbd.set_velocity_(state, rbd.rand_velocity(model, B, rng))
shp = (lambda n: (B, n)) if layout == "aos" else (lambda n: (n, B))
vd = torch.rand(shp(model.nv), dtype=tdt, device="cuda"); out = torch.zeros_like(vd)
jw = torch.zeros(shp(6 * model.n_bodies), dtype=tdt, device="cuda"); acc = torch.zeros_like(jw)
for _ in range(40): rbd.inverse_dynamics_(out, state, vd, jointwrenchesout=jw, accelerations=acc)
torch.cuda.synchronize()
print(layout, rbd.last_kernel(state))
