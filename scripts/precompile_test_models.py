"""Compiles the run-time specialised kernels of every mechanism the GPU tests use into the library's cache (csrc/jit_cache/, no GPU needed), so that the
GPU box only loads them.  Optional: anything missing is compiled there on first use."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import rbd_amd as rbd
names = ["atlas_floating", "atlas_fixed", "valkyrie_floating", "double_pendulum", "acrobot_urdf", "quickstart_pendulum", "randmech2", "randmech3"]
for name in names:
    path = os.path.join(ROOT, "tests", "golden", "models", name + ".json")
    if not os.path.exists(path):
        continue
    model = rbd.load_flat_model(path)
    for dt in (torch.float32, torch.float64):
        t = time.time()
        ok = rbd.jit_precompile(model, dt)[0]
        print(name, dt, ok, round(time.time() - t, 1), "s", flush=True)
