"""Host-side invariants of the chain schedule under the track / walk plans (rbd_model_chain_plan; csrc/rbd_chain_plan.hpp).  No GPU needed: the
plan is index bookkeeping built by rbd_model_create."""
import numpy as np
import pytest

PLANNED = ["atlas_floating", "atlas_fixed", "valkyrie_floating", "double_pendulum", "acrobot_urdf"]


def check_plan(flat, plan):
    tab = plan["table"]
    nb = flat.n_bodies
    # every body exactly once
    placed = tab[tab >= 0]
    assert sorted(placed.tolist()) == list(range(nb))
    step = {int(b): s for s in range(tab.shape[0]) for b in tab[s] if b >= 0}
    # a body runs after its parent (pass A / C order; pass B is the reverse)
    for b in range(nb):
        p = int(flat.parent[b])
        if p >= 0:
            assert step[b] > step[p]
    # the schedule is no longer than one body per step, and at least the depth of the tree
    depth = np.zeros(nb, int)
    for b in range(nb):
        depth[b] = 0 if flat.parent[b] < 0 else depth[flat.parent[b]] + 1
    assert depth.max() + 1 <= plan["steps"] <= nb
    assert plan["tracks"] in (1, 2, 4, 8, 16)


@pytest.mark.parametrize("name", PLANNED)
def test_chain_plan_invariants(rbd, models, name):
    flat = models[name]
    plan = rbd.chain_plan(flat)
    assert plan is not None
    check_plan(flat, plan)


def test_chain_plan_atlas_reaches_the_critical_path(rbd, models):
    """Atlas: 5 chains on 4 tracks; the schedule is as long as the deepest chain (pelvis ... hand: 11 bodies), and the LDS
    footprint (fields x 16 states x 8 B) lets three fp64 wavefronts share a CU."""
    plan = rbd.chain_plan(models["atlas_floating"])
    assert plan["tracks"] == 4 and plan["steps"] == 11


@pytest.mark.parametrize("name", ["randmech1", "randmech2", "randmech3", "inner_floating"])
def test_chain_plan_scope(rbd, models, name):
    """3-dof tree joints and 6-dof joints below another body stay with the lane-per-body mapping; anything else plans."""
    flat = models[name]
    jt = np.asarray(flat.joint_type)
    out_of_scope = np.any((jt == 4) | (jt == 5)) or np.any((jt == 3) & (np.asarray(flat.parent) >= 0))
    plan = rbd.chain_plan(flat)
    assert (plan is None) == bool(out_of_scope)
    if plan is not None:
        check_plan(flat, plan)


def random_tree(rbd, rng, n, floating_root, chain_bias):
    """rand_tree_mechanism (src/mechanism_modification.jl:382-396) with a parent selector that mixes chains and bushes."""
    types = (["QuaternionFloating"] if floating_root else []) + [str(rng.choice(["Revolute", "Prismatic", "Fixed", "SinCosRevolute"])) for _ in range(n)]

    def selector(mech, r):
        bodies = [b for b in mech.bodies if not (floating_root and b is mech.bodies[0])]  # keep the 6-dof joint the only child of the world
        return bodies[-1] if r.random() < chain_bias else bodies[r.integers(len(bodies))]

    return rbd.rand_tree_mechanism(rng, types, selector)


def test_chain_plan_random_trees(rbd):
    """Random topologies (revolute / prismatic / fixed / sin-cos joints, up to 40 bodies, from chain-like to bushy)."""
    rng = np.random.default_rng(5)
    for trial in range(20):
        mech = random_tree(rbd, rng, int(rng.integers(1, 40)), bool(trial % 2), float(rng.uniform(0, 1)))
        flat = rbd.flatten(mech)
        plan = rbd.chain_plan(flat)
        assert plan is not None
        check_plan(flat, plan)


def test_bank_plan(rbd, models):
    """Two bodies per lane: Atlas splits at level 5 into 15 + 16 bodies on 16 lanes (4 states per wavefront instead of 2)."""
    plan = rbd.bank_plan(models["atlas_floating"])
    assert plan == {"lanes": 16, "L0": 5, "bodies": (15, 16), "aba": True}
    assert rbd.bank_plan(models["randmech1"])["aba"] is False  # 3-dof joints: banked RNEA only
    rng = np.random.default_rng(6)
    for trial in range(30):
        flat = rbd.flatten(random_tree(rbd, rng, int(rng.integers(1, 40)), bool(trial % 2), float(rng.uniform(0, 1))))
        plan = rbd.bank_plan(flat)
        lps = 1
        while lps < flat.n_bodies:
            lps <<= 1
        if plan is None:
            continue
        lv = flat.levels()
        n0, n1 = int((lv < plan["L0"]).sum()), int((lv >= plan["L0"]).sum())
        assert plan["bodies"] == (n0, n1) and n0 + n1 == flat.n_bodies
        assert max(n0, n1) <= plan["lanes"] < lps  # fits, and saves lanes over one body per lane
