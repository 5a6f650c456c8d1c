"""Generates tests/golden/models/*.json — flattened mechanisms (rbd_flat_model_t images) so that tests,
smoke() and bench.py can run where /root/reference does not exist (the GPU box).

Run in the build container:  python tests/golden/make_models.py
Sources: /root/reference/test/urdf/atlas.urdf (the vendored Atlas v5 fixture used instead of the
benchmark's downloaded atlas_minimal_contact.urdf — SURVEY.md F7), Acrobot.urdf and valkyrie.urdf,
parsed with rigidbodydynamics.jl_amd/urdf.py (parse_urdf semantics: src/urdf/parse.jl:162-221).
tests/test_golden_vectors.py::test_model_fixtures_regenerate_from_the_reference_urdfs re-runs build_all() and compares every array
with the committed files."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/test/urdf"
OUT = os.path.join(HERE, "models")

SPECS = {
    "atlas_floating": ("atlas.urdf", dict(floating=True), {"source": "test/urdf/atlas.urdf", "floating": True}),
    "atlas_fixed": ("atlas.urdf", dict(floating=False), {"source": "test/urdf/atlas.urdf", "floating": False}),
    "acrobot_urdf": ("Acrobot.urdf", dict(remove_fixed_tree_joints=False), {"source": "test/urdf/Acrobot.urdf", "remove_fixed_tree_joints": False}),
    "valkyrie_floating": ("valkyrie.urdf", dict(floating=True), {"source": "test/urdf/valkyrie.urdf", "floating": True}),
}


def build_all(rbd):
    """name -> FlatModel, parsed afresh from the reference's URDF fixtures."""
    return {name: rbd.flatten(rbd.parse_urdf(f"{REF}/{urdf}", **kw)) for name, (urdf, kw, _) in SPECS.items()}


def main():
    sys.path.insert(0, os.path.join(HERE, "..", ".."))
    import rbd_amd as rbd
    os.makedirs(OUT, exist_ok=True)
    for name, fm in build_all(rbd).items():
        rbd.save_flat_model(fm, os.path.join(OUT, name + ".json"), SPECS[name][2])
        print(name, fm.n_bodies, fm.nq, fm.nv, f"{fm.total_mass():.6f}")


if __name__ == "__main__":
    main()
