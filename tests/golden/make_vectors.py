"""Writes tests/golden/vectors/*.npz — committed known-answer vectors for the hot path.

The reference (pure Julia) cannot run in the build image and stores no vectors itself, so these are the next best pin:
for every fixture mechanism, seeded inputs (q, v, τ / v̇, w_ext drawn with the reference's distributions) and the outputs
    tau  = inverse_dynamics(q, v, v̇, w_ext)      c = dynamics_bias(q, v, w_ext)      M = mass_matrix(q)      vdot = dynamics(q, v, τ, w_ext)
computed by oracle/featherstone_np.py — the textbook body-coordinate formulation, independent of the C restatement the GPU tests use as
their checker.  tests/test_golden_vectors.py then requires the C oracle AND the GPU to reproduce them (1e-11 / 1e-10), and requires this
script to regenerate the committed files (so neither implementation can drift unnoticed).  The four-bar linkage (loop joints) has no
second implementation: its vectors come from the C oracle's constrained route and are marked `source = "rbd_oracle (single source)"`.

usage: python tests/golden/make_vectors.py [--check]
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
B = 6


def fixture_models(rbd):
    m = {name: rbd.load_flat_model(os.path.join(HERE, "models", name + ".json")) for name in ("atlas_floating", "atlas_fixed", "valkyrie_floating", "acrobot_urdf")}
    m["double_pendulum"] = rbd.flatten(rbd.double_pendulum())
    m["randmech1"] = rbd.flatten(rbd.randmech(np.random.default_rng(1)))
    m["inner_floating"] = rbd.flatten(rbd.rand_tree_mechanism(np.random.default_rng(7), ["Revolute", "Prismatic", "QuaternionFloating", "Revolute", "QuaternionSpherical", "Planar", "QuaternionFloating", "Revolute"]))
    m["four_bar"] = rbd.flatten(rbd.four_bar_linkage())
    return m


def vectors_for(rbd, name, model):
    import featherstone_np as fs
    rng = np.random.default_rng(abs(hash(name)) % 1000 if False else sum(map(ord, name)))
    if name == "four_bar":
        import oracle
        q = np.tile(np.asarray(rbd.FOUR_BAR_INITIAL_Q, float), (B, 1)); q[:, 0] += rng.uniform(-0.05, 0.05, B)
        v = np.tile(np.asarray(rbd.FOUR_BAR_INITIAL_V, float), (B, 1)) * rng.uniform(0.5, 1.5, (B, 1))
        tau = rng.random((B, model.nv))
        r = oracle.dynamics_loops(model, q, v, tau, None, stabilize=True)
        return dict(q=q, v=v, tau_in=tau, vdot=r["vdot"], lam=r["lam"], K=r["K"], k=r["k"], M=r["M"], c=r["c"], source=np.array("rbd_oracle (single source)"))
    q, v = rbd.rand_configuration(model, B, rng), rbd.rand_velocity(model, B, rng)
    tau_in, vd_in, fext = rng.random((B, model.nv)), rng.random((B, model.nv)), rng.random((B, 6 * model.n_bodies))
    return dict(q=q, v=v, tau_in=tau_in, vd_in=vd_in, fext=fext,
                tau=fs.batch(fs.rnea, model, q, v, vd_in, fext), c=fs.batch(fs.rnea, model, q, v, None, fext), c_nowrench=fs.batch(fs.rnea, model, q, v, None, None),
                M=fs.batch(fs.crba, model, q), vdot=fs.batch(fs.aba, model, q, v, tau_in, fext), vdot_nowrench=fs.batch(fs.aba, model, q, v, tau_in, None),
                source=np.array("featherstone_np (RBDA Tables 5.1 / 6.2 / 7.1)"))


def main():
    import rbd_amd as rbd
    check = "--check" in sys.argv
    out_dir = os.path.join(HERE, "vectors")
    os.makedirs(out_dir, exist_ok=True)
    worst = 0.0
    for name, model in fixture_models(rbd).items():
        vec = vectors_for(rbd, name, model)
        path = os.path.join(out_dir, name + ".npz")
        if check:
            old = np.load(path)
            for k, a in vec.items():
                if a.dtype.kind == "f":
                    worst = max(worst, float(np.abs(a - old[k]).max() / max(1.0, np.abs(old[k]).max())))
        else:
            np.savez(path, **vec)
    print("regenerated vectors differ from the committed ones by", worst) if check else print("written", out_dir)
    return worst


if __name__ == "__main__":
    main()
