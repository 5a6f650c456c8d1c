"""Compiles the run-time specialised kernels of every mechanism the GPU tests use into the library's cache (csrc/jit_cache/, no GPU needed), so that the
GPU box only loads them.  Optional: anything missing is compiled there on first use."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import rbd_amd as rbd
import numpy as np
models = {name: rbd.load_flat_model(os.path.join(ROOT, "tests", "golden", "models", name + ".json")) for name in ("atlas_floating", "atlas_fixed", "valkyrie_floating", "acrobot_urdf")}
models["double_pendulum"] = rbd.flatten(rbd.double_pendulum())          # (as tests/conftest.py builds them)
models["quickstart_pendulum"] = rbd.flatten(rbd.quickstart_double_pendulum())
models["four_bar"] = rbd.flatten(rbd.four_bar_linkage())
for seed in (1, 2, 3):
    models[f"randmech{seed}"] = rbd.flatten(rbd.randmech(np.random.default_rng(seed)))
models["inner_floating"] = rbd.flatten(rbd.rand_tree_mechanism(np.random.default_rng(7), ["Revolute", "Prismatic", "QuaternionFloating", "Revolute", "QuaternionSpherical", "Planar", "QuaternionFloating", "Revolute"]))
models["mixed20"] = rbd.flatten(rbd.rand_tree_mechanism(np.random.default_rng(11), ["QuaternionFloating", "QuaternionSpherical", "Planar", "Revolute", "Revolute", "Prismatic",
                                                                                   "QuaternionSpherical", "Revolute", "SinCosRevolute"]))
for name, model in models.items():
    for dt in (torch.float32, torch.float64):
        t = time.time()
        ok, log = rbd.jit_precompile(model, dt)
        refused = [l for l in log.splitlines() if "not used" in l]  # (a walk program the allocator gave accumulation registers or scratch: the interpreting kernel serves instead)
        print(name, dt, ok, round(time.time() - t, 1), "s", ("  REFUSED: " + " | ".join(refused)) if refused else "", flush=True)
