"""Generates tests/golden/models/*.json — flattened mechanisms (rbd_flat_model_t images) so that tests,
smoke() and bench.py can run where /root/reference does not exist (the GPU box).

Run in the build container:  python tests/golden/make_models.py
Sources: /root/reference/test/urdf/atlas.urdf (the vendored Atlas v5 fixture used instead of the
benchmark's downloaded atlas_minimal_contact.urdf — SURVEY.md F7) and /root/reference/test/urdf/Acrobot.urdf,
parsed with rigidbodydynamics.jl_amd/urdf.py (parse_urdf semantics: src/urdf/parse.jl:162-221)."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import rbd_amd as rbd  # noqa: E402

REF = "/root/reference/test/urdf"
OUT = os.path.join(HERE, "models")
os.makedirs(OUT, exist_ok=True)


def dump(name, mech, meta):
    fm = rbd.flatten(mech)
    rbd.save_flat_model(fm, os.path.join(OUT, name + ".json"), meta)
    print(name, fm.n_bodies, fm.nq, fm.nv, f"{fm.total_mass():.6f}")


dump("atlas_floating", rbd.parse_urdf(f"{REF}/atlas.urdf", floating=True), {"source": "test/urdf/atlas.urdf", "floating": True})
dump("atlas_fixed", rbd.parse_urdf(f"{REF}/atlas.urdf", floating=False), {"source": "test/urdf/atlas.urdf", "floating": False})
dump("acrobot_urdf", rbd.parse_urdf(f"{REF}/Acrobot.urdf", remove_fixed_tree_joints=False),
     {"source": "test/urdf/Acrobot.urdf", "remove_fixed_tree_joints": False})
dump("valkyrie_floating", rbd.parse_urdf(f"{REF}/valkyrie.urdf", floating=True), {"source": "test/urdf/valkyrie.urdf", "floating": True})
