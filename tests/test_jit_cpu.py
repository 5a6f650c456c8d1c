"""Run-time specialisation, host side (csrc/rbd_jit.hip, rbd_spec.hpp): the generated source of a mechanism's kernels — plan tables, the
factorisation order, the tile mask, the gather lists of the M emitter — checked against the mechanism's own structure, and its compilation by
hiprtc, which needs no GPU.  What the compiled kernels compute is checked on the GPU (tests/test_state_kernels.py)."""
import os
import re
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def rbd():
    import rbd_amd
    return rbd_amd


def table(src, name):
    m = re.search(r"\b%s(?:\[[^\]]*\])+ = \{(.*?)\};" % re.escape(name), src, re.S)
    assert m, name
    rows = [r for r in re.findall(r"\{([^{}]*)\}", m.group(1))]
    if not rows:
        rows = [m.group(1)]
    return [[int(x, 0) if not re.search(r"[.eE]|inf|nan", x) or x.strip().startswith("0x") else float(x) for x in re.split(r",", r.replace("ull", "")) if x.strip()] for r in rows]


def ancestors(model):
    """anc[i][j]: velocity coordinate j belongs to a joint on the path from coordinate i's joint to the root (or to the same joint)."""
    nb, nv = model.n_bodies, model.nv
    nvs = [(int(model.v_offset[b + 1]) if b + 1 < nb else nv) - int(model.v_offset[b]) for b in range(nb)]
    body_of = np.concatenate([[b] * nvs[b] for b in range(nb)]).astype(int) if nv else np.zeros(0, int)
    anc = np.zeros((nv, nv), bool)
    for i in range(nv):
        b = body_of[i]
        while b >= 0:
            anc[i, body_of == b] = True
            b = int(model.parent[b])
    return anc


def mechanism(rbd, name):
    """the URDF fixtures, and the in-code mechanisms with 3-dof joints / 6-dof joints below the world as tests/conftest.py builds them"""
    if name == "mixed20":  # nv = 20: the Cholesky compiled for the sparsity applies
        return rbd.flatten(rbd.rand_tree_mechanism(np.random.default_rng(11), ["QuaternionFloating", "QuaternionSpherical", "Planar", "Revolute", "Revolute", "Prismatic",
                                                                                "QuaternionSpherical", "Revolute", "SinCosRevolute"]))
    if name == "inner_floating":
        return rbd.flatten(rbd.rand_tree_mechanism(np.random.default_rng(7), ["Revolute", "Prismatic", "QuaternionFloating", "Revolute", "QuaternionSpherical", "Planar", "QuaternionFloating", "Revolute"]))
    if name.startswith("randmech"):
        return rbd.flatten(rbd.randmech(np.random.default_rng(int(name[8:]))))
    return rbd.load_flat_model(os.path.join(ROOT, "tests", "golden", "models", name + ".json"))


@pytest.mark.parametrize("name", ["atlas_floating", "valkyrie_floating", "atlas_fixed", "mixed20", "inner_floating", "randmech1"])
def test_generated_plan_matches_the_mechanism(rbd, name):
    model = mechanism(rbd, name)
    src = rbd.jit_source(model, torch.float32)
    assert src is not None
    dims = re.search(r"constexpr int NB = (\d+), NQ = (\d+), NV = (\d+), NOPS = (\d+), NLEVELS = (\d+);", src)
    nb, nq, nv, nops, nlev = map(int, dims.groups())
    # fp32 programs walk sibling limbs of the same shape in lockstep (rbd_jit.hip: merge_limbs): an op of a limb stands for two bodies, the partner's words in OPW2
    npair = int(re.search(r"constexpr int NPAIR = (\d+);", src).group(1))
    assert (nb, nq, nv, nops) == (model.n_bodies, model.nq, model.nv, 2 * (model.n_bodies - npair))
    opw, opw2, pair = table(src, "OPW"), table(src, "OPW2"), table(src, "PAIR")[0]
    assert len(opw) == len(opw2) == len(pair) == nops and sum(pair) == 2 * npair
    enter = [w for w in opw if (w[0] & 0xff) == 0] + [w2 for w, w2, p_ in zip(opw, opw2, pair) if p_ and (w[0] & 0xff) == 0]
    assert all(w2[0] == w[0] for w, w2, p_ in zip(opw, opw2, pair) if p_)  # partners: same kind, level and joint type
    assert sorted(w[2] for w in enter) == sorted(int(v) for v in model.v_offset)  # every body entered once
    anc = ancestors(model)
    nz = anc | anc.T  # support of the mass matrix (src/mechanism_state.jl:95-98)
    rowmask = table(src, "ROWMASK")[0]
    for i in range(nv):
        for j in range(i + 1):
            assert bool((rowmask[i] >> j) & 1) == bool(nz[i, j])
    if "RBD_SPEC_CHOL" not in src:
        assert nv % 4 or nv > 40
        return
    perm, inv = np.array(table(src, "PERM")[0]), np.array(table(src, "INV")[0])
    assert sorted(perm) == list(range(nv)) and (inv[perm] == np.arange(nv)).all()
    # children before parents: a proper ancestor's coordinate comes later in the factorisation order -> no fill-in
    for i in range(nv):
        for j in range(nv):
            if anc[i, j] and not anc[j, i]:
                assert perm[j] > perm[i]
    nt = nv // 4
    tmask = np.array(table(src, "TMASK"))
    want = np.zeros((nt, nt), int)
    for pr in range(nv):
        for pc in range(pr + 1):
            if nz[inv[pr], inv[pc]]:
                want[pr // 4, pc // 4] = 1
    assert (tmask == want).all()
    # no fill-in, tile by tile: whenever two tiles of a block column are in the mask and hold a common descendant, their product's target is too
    L = np.zeros((nv, nv), bool)
    for pr in range(nv):
        for pc in range(pr + 1):
            L[pr, pc] = nz[inv[pr], inv[pc]]
    for k in range(nv):  # symbolic Cholesky
        rows = [i for i in range(k + 1, nv) if L[i, k]]
        for a in rows:
            for b_ in rows:
                if a >= b_:
                    assert L[a, b_], "fill-in at (%d, %d)" % (a, b_)
    # the emitter's gather lists: every non-zero of the full square exactly once, from the right staged entry, to the right slot
    emit = table(src, "EMIT")
    ks = table(src, "EMIT_K")[0]
    seen = set()
    for jo in range(nt):
        for e in emit[jo][:4 * ks[jo]]:
            srce, dst = e >> 16, e & 0xffff
            if dst == 4 * nv:
                continue  # padding
            col, row = 4 * jo + dst // nv, dst % nv
            hi, lo = max(perm[row], perm[col]), min(perm[row], perm[col])
            assert srce == lo * nv + hi and nz[row, col] and (row, col) not in seen
            seen.add((row, col))
        assert all((e & 0xffff) == 4 * nv for e in emit[jo][4 * ks[jo]:])
    assert len(seen) == int(nz.sum())


def test_programs_of_mechanisms_with_every_joint_type(rbd, tmp_path, monkeypatch):
    """Planar / QuaternionSpherical joints and QuaternionFloating joints below the world (the reference's randmech, test/test_mechanism_algorithms.jl:1-11): all three
    one-lane-per-state programs exist, the ancestor columns carry the joint's kind, the 3-dof joints are ranked (dynamics! keeps part of U D^-1 in rows of their
    own), and the programs compile without a device."""
    model = mechanism(rbd, "inner_floating")
    jt = [int(t) for t in model.joint_type]
    n3 = sum(1 for t in jt if t in (4, 5))
    assert n3 == 2 and jt.count(3) == 2
    for fam in ("mass_matrix", "dynamics", "inverse_dynamics"):
        for dt in (torch.float32, torch.float64):
            src = rbd.jit_source(model, dt, fam)
            assert src is not None, (fam, dt)  # (round 6: dynamics! in fp64 too — for exactly these mechanisms; an Atlas-like tree has none)
    atlas = mechanism(rbd, "atlas_floating")
    assert rbd.jit_source(atlas, torch.float64, "dynamics") is None and rbd.jit_source(atlas, torch.float32, "dynamics") is not None
    # fp64: two programs (csrc/rbd_spec.hpp aba_spec UDL, GST) — every row in LDS (one CU's 160 KB: one wavefront per CU), or the spare and 3-dof rows in an HBM
    # stash (half of it: two) — each with as many of a body's four register values of U D^-1 in LDS rows as its budget has room for, at most three
    s64 = rbd.jit_source(model, torch.float64, "dynamics")
    for gst, kb in ((0, 160), (1, 80)):
        rows = model.nq + 2 * model.nv + (0 if gst else max(model.nq, model.n_bodies) + 10 * n3)
        udl = max(0, min(3, (kb * 1024 // (65 * 8) - rows) // model.n_bodies))
        for fx in ("", "nofext_"):
            body = s64[s64.index("void aba_spec_%s%sf64(" % ("gst_" if gst else "", fx)):]
            body = body[:body.index("}\n") + 1]
            assert "aba_lds[%d]" % ((rows + udl * model.n_bodies) * 65) in body and "aba_spec<double, %s, %d, %d>" % ("false" if fx else "true", udl, gst) in body, body
    assert "NPAIR = 0" in s64
    src = rbd.jit_source(model, torch.float32, "dynamics")
    assert "aba_spec_gst" not in src and "aba_spec<float, true, 0, 0>" in src
    assert int(re.search(r"N3 = (\d+);", src).group(1)) == n3
    opw, x3, cols = table(src, "OPW"), table(src, "X3")[0], table(src, "COLS")
    for o, w in enumerate(opw):
        t = w[0] >> 16
        assert (x3[o] >= 0) == (t in (4, 5))
    assert sorted({x for x in x3 if x >= 0}) == list(range(n3))
    flag = {3: 0x10000, 5: 0x20000, 4: 0x40000}
    body_of_voff = {int(model.v_offset[b]): b for b in range(model.n_bodies)}
    seen = set()
    for row in cols:
        for c in row:
            if c >= 0:
                b = body_of_voff[c & 0xffff]
                assert (c & 0x70000) == flag.get(jt[b], 0), (c, jt[b])
                seen.add(jt[b])
    assert {3, 4, 5} & seen  # a joint with several coordinates is somebody's ancestor in this mechanism
    monkeypatch.setenv("RBD_JIT_CACHE", str(tmp_path / "cache"))
    ok, log = rbd.jit_precompile(model, torch.float32)
    if ok is None:
        pytest.skip("hiprtc not available")
    assert ok, log


def test_a_process_that_exits_waits_for_its_background_compilations(tmp_path):
    """A process that ends while kernels are still being compiled for it: the `atexit` hook of the Python mirror (rbd_jit_wait_idle, include/rbd_hip.h) waits for the
    compiler — without it the compiler's own teardown pulled the ground from under its thread (segmentation fault at exit) — and the cache has the programs."""
    import subprocess, sys
    cache = tmp_path / "cache"
    code = (
        "import os, sys\n"
        "os.environ['RBD_JIT_CACHE'] = %r\n"
        "sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'oracle'))\n"
        "import torch, rbd_amd as rbd\n"
        "m = rbd.flatten(rbd.builders.double_pendulum())\n"
        "st = [rbd.jit_status(m, torch.float32, f) for f in (0, 1, 2)]\n"
        "print('status', st)\n" % (str(cache), ROOT, ROOT))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=280)
    if "status [-1, -1, -1]" in r.stdout:
        pytest.skip("hiprtc not available")
    assert r.returncode == 0, (r.returncode, r.stdout[-500:], r.stderr[-2000:])
    assert "status [0, 0, 0]" in r.stdout, r.stdout  # all three were started in the background, none waited for
    files = [f for f in os.listdir(cache) if f.endswith(".hsaco")]
    assert len(files) == 3 and not [f for f in os.listdir(cache) if ".tmp" in f]


def test_out_of_scope_mechanisms_have_no_specialised_kernels(rbd):
    mech = rbd.builders.four_bar_linkage()  # loop joints: the state plan does not apply
    assert rbd.jit_source(rbd.flatten(mech), torch.float32) is None


def test_precompile_without_a_device(rbd, tmp_path, monkeypatch):
    """hiprtc cross-compiles for gfx950 with no GPU present: the code object lands in the cache, a second call finds it there."""
    monkeypatch.setenv("RBD_JIT_CACHE", str(tmp_path))
    model = rbd.flatten(rbd.builders.double_pendulum())
    ok, log = rbd.jit_precompile(model, torch.float64)
    if ok is None:
        pytest.skip("libhiprtc not available")
    assert ok, log
    files = sorted(f for f in os.listdir(tmp_path) if f.endswith(".hsaco"))
    # fp64: the mass-matrix and the inverse-dynamics programs, and the two walk kernels' (each compiled ONCE since round 4: the allocator's register use is read from
    # the object's metadata and its kernel descriptor rewritten to cover the accumulation registers, csrc/rbd_jit.hip jit_kd_cover_agprs)
    # ... and, where the mechanism has a two-bodies-per-lane split, the banked kernels' (round 4: their level loops unrolled against the mechanism's level structure)
    # ... and (round 5) the walk kernel that takes the four stages of a `simulate` step in one launch, a program of its own
    # ... and (round 6) the kinematics by-products' five kernels, one program (family 11)
    n = 6 + (rbd.jit_source(model, torch.float64, "banked") is not None)
    assert len(files) == n and all(os.path.getsize(tmp_path / f) > 1000 for f in files)
    assert "ready after" in log and log.count("[rbd_jit] family") == n  # the log lists every program with the seconds it took
    stamps = [os.path.getmtime(tmp_path / f) for f in files]
    ok, _ = rbd.jit_precompile(model, torch.float64)
    assert ok and [os.path.getmtime(tmp_path / f) for f in files] == stamps
    monkeypatch.setenv("RBD_JIT", "0")
    assert rbd.jit_precompile(model, torch.float64)[0] is None


def test_compilation_in_the_background(rbd, tmp_path, monkeypatch):
    """A program that is not in the cache is compiled on a background thread: rbd_jit_status — the query the hot-path calls make before they pick a kernel
    (spec_load / spec_walk / spec_loop in csrc/rbd_capi.hip) — returns at once with 0 while hiprtc runs, and 1 once the code object is there; the object
    is in the cache afterwards, so a second process finds it ready.  No device needed."""
    import time
    monkeypatch.setenv("RBD_JIT_CACHE", str(tmp_path / "cache"))  # (a missing directory is created, 0700)
    model = rbd.flatten(rbd.builders.four_bar_linkage())
    if rbd.jit_precompile(rbd.flatten(rbd.builders.double_pendulum()), torch.float32)[0] is None:
        pytest.skip("libhiprtc not available")
    LOOP = 3  # family 3: the whole loop branch of a small loop mechanism
    assert rbd.jit_status(model, torch.float32, 0) == -1  # no one-lane-per-state programs for a mechanism with loop joints
    t0 = time.time()
    first = rbd.jit_status(model, torch.float64, LOOP)
    dt_first = time.time() - t0
    assert first == 0 and dt_first < 0.5, (first, dt_first)  # started, not waited for (the compilation takes 1.5-3 s)
    assert rbd.jit_status(model, torch.float64, LOOP) in (0, 1)  # asking again does not start a second compilation: still one object at the end
    deadline = time.time() + 120
    while rbd.jit_status(model, torch.float64, LOOP) == 0:
        assert time.time() < deadline, "background compilation never finished"
        time.sleep(0.05)
    assert rbd.jit_status(model, torch.float64, LOOP) == 1
    cache = tmp_path / "cache"
    files = [f for f in os.listdir(cache) if f.endswith(".hsaco")]
    assert len([f for f in files if os.path.getsize(cache / f) > 1000]) >= 1 and not [f for f in os.listdir(cache) if ".tmp" in f]
    assert (os.stat(cache).st_mode & 0o077) == 0
    # a cache directory somebody else could write to is not used (a code object is loaded by name alone): caching is off, compilation still works
    shared = tmp_path / "shared"
    shared.mkdir()
    os.chmod(shared, 0o777)
    monkeypatch.setenv("RBD_JIT_CACHE", str(shared))
    ok, _ = rbd.jit_precompile(rbd.flatten(rbd.builders.double_pendulum()), torch.float32)
    assert ok and not os.listdir(shared)


@pytest.mark.parametrize("name", ["atlas_floating", "valkyrie_floating", "atlas_fixed"])
def test_generated_tree_tables_of_the_walk_kernels(rbd, name):
    """The dynamics! / inverse_dynamics! programs: children counts, sibling ranks and branch slots of every op's body against the mechanism."""
    model = rbd.load_flat_model(os.path.join(ROOT, "tests", "golden", "models", name + ".json"))
    for fam in ("dynamics", "inverse_dynamics"):
        src = rbd.jit_source(model, torch.float32, fam)
        assert src is not None and "RBD_SPEC_ABA" in src
        assert (rbd.jit_source(model, torch.float64, fam) is None) == (fam == "dynamics")  # fp64: inverse dynamics only
        opw = table(src, "OPW")
        T = {k: table(src, k)[0] for k in ("BODY", "NCH", "BS", "PBS", "CIDX", "NEXT_EXIT")}
        nbs = int(re.search(r"constexpr int NBS = (\d+), FIRST_EXIT = (\d+), N3 = \d+;", src).group(1))
        first_exit = int(re.search(r"FIRST_EXIT = (\d+),", src).group(1))
        voff_to_body = {int(v): b for b, v in enumerate(model.v_offset)}
        nvs = [(int(model.v_offset[b + 1]) if b + 1 < model.n_bodies else model.nv) - int(model.v_offset[b]) for b in range(model.n_bodies)]
        children = {b: [c for c in range(model.n_bodies) if model.parent[c] == b] for b in range(-1, model.n_bodies)}
        stack, seen_rank, exits = [], {}, []
        for o, w in enumerate(opw):
            kind, lvl = w[0] & 0xff, (w[0] >> 8) & 0xff
            if kind == 0:
                assert lvl == len(stack)
                # which body: by its v offset when it has coordinates (fixed joints: by elimination through the children counts below)
                stack.append(o)
                par = stack[-2] if len(stack) > 1 else None
                if par is not None:
                    seen_rank.setdefault(par, []).append(T["CIDX"][o])
                    assert (T["PBS"][o] >= 0) == (T["NCH"][par] >= 2) and (T["PBS"][o] < 0 or T["PBS"][o] == T["BS"][par])
                assert (T["BS"][o] >= 0) == (T["NCH"][o] >= 2)
                if T["BS"][o] >= 0:
                    assert T["BS"][o] == sum(1 for a in stack[:-1] if T["NCH"][a] >= 2) < nbs
            else:
                e = stack.pop()
                exits.append(o)
                for k in ("BODY", "NCH", "BS", "PBS", "CIDX"):
                    assert T[k][o] == T[k][e]
                assert sorted(seen_rank.get(e, [])) == list(range(T["NCH"][e]))
        assert not stack and exits[0] == first_exit
        # limbs in lockstep: an op of a limb stands for two bodies (PAIR), the partner's ordinal in BODY2; a limb's first op (PROOT) hangs off a body walked alone
        P2 = {k: table(src, k)[0] for k in ("PAIR", "PROOT", "BODY2")}
        ent = [o for o, w in enumerate(opw) if (w[0] & 0xff) == 0]
        assert sorted([T["BODY"][o] for o in ent] + [P2["BODY2"][o] for o in ent if P2["PAIR"][o]]) == list(range(model.n_bodies))
        assert name != "atlas_floating" or sum(P2["PAIR"][o] for o in ent) == 13  # Atlas: two arms of seven bodies, two legs of six
        for a, b in zip(exits, exits[1:] + [-1]):
            assert T["NEXT_EXIT"][a] == b
        # children counts against the mechanism, body by body (bodies with coordinates are identified by their v offset)
        # (a pair of limbs counts as ONE child of the body they hang off)
        limbs_below = {o: 0 for o in ent}
        stack = []
        for o, w in enumerate(opw):
            if (w[0] & 0xff) == 0:
                if stack and P2["PROOT"][o]:
                    limbs_below[stack[-1]] += 1
                stack.append(o)
            else:
                stack.pop()
        for o, w in enumerate(opw):
            if (w[0] & 0xff) == 0 and (w[0] >> 16) != 0 and w[2] in voff_to_body and nvs[voff_to_body[w[2]]] > 0:  # 0 = RBD_JOINT_FIXED
                assert T["NCH"][o] + limbs_below[o] == len(children[voff_to_body[w[2]]])


def walk_trees(rbd):
    """The seeded random trees of tests/test_state_kernels.py::test_compiled_walk_random_trees (compiled here, on the CPU, they travel to the GPU box in the cache)."""
    from test_chain_plan import random_tree
    rng = np.random.default_rng(41)
    return [rbd.flatten(random_tree(rbd, rng, n, floating, 0.4)) for n, floating in ((5, False), (7, True))]


def test_walk_program_of_a_mechanism(rbd):
    """The one-wavefront-per-track dynamics! kernel compiled per mechanism (aba_walk_spec, csrc/rbd_walk.hpp): the plan's records, constants and
    parking words as tables of ns x G entries, the barrier masks, the re-rooted tree's chain; the rows of 64 states as static LDS."""
    model = rbd.load_flat_model(os.path.join(ROOT, "tests", "golden", "models", "atlas_floating.json"))
    # fp32: one state per lane and two (packed arithmetic; the rows are 8 bytes wide like fp64's); the pair programs exist in fp32 only
    s32, s32x2 = rbd.jit_source(model, torch.float32, "dynamics_tracks"), rbd.jit_source(model, torch.float32, "dynamics_tracks_pairs")
    assert "aba_walk_spec_f32(" in s32 and "rbd::aba_walk_spec<float," in s32 and "aba_walk_spec_f32x2(" in s32x2 and "rbd::aba_walk_spec<rbd::f2," in s32x2
    assert 2 * int(re.search(r"unsigned char lds\[(\d+)\]", s32).group(1)) == int(re.search(r"unsigned char lds\[(\d+)\]", s32x2).group(1))
    assert rbd.jit_source(model, torch.float64, "dynamics_tracks_pairs") is None
    assert "rnea_walk_spec_f32x2(" in rbd.jit_source(model, torch.float32, "inverse_dynamics_tracks_pairs")
    src = rbd.jit_source(model, torch.float64, "dynamics_tracks")
    assert src is not None and "aba_walk_spec_f64" in src and '#include "rbd_walk.hpp"' in src
    ns, G, nq, nv, n1, nf, fw = (int(x) for x in re.search(r"NS = (\d+), G = (\d+), NQ = (\d+), NV = (\d+), MK_N1 = (\d+), MK_NF = (\d+), MK_FW = (\d+);", src).groups())
    assert 0 <= fw < G  # (the wavefront with the shortest track: it takes the 6-dof joints' stage arithmetic)
    # the `simulate` program: the same plan, the instantiation with the passes inside a loop over the four stages, machine-level LICM off, and the marker that lets
    # its register allocator take accumulation registers of its own — below the stash's lowest register, which the marker names (jit_walk_admit)
    sim = rbd.jit_source(model, torch.float64, "dynamics_tracks_sim")
    assert "aba_walk_sim_spec_f64(" in sim and "rbd_walk_tables::Plan, true>" in sim and "// rbd-walk-sim-loop stash-from=a%d\n" % (256 - 18 * ns) in sim and "-disable-machine-licm" in sim
    assert "rbd_walk_tables::Plan, false>" in src and "rbd-walk-sim-loop" not in src and "disable-machine-licm" not in src
    assert "aba_walk_sim_spec_f32x2(" in rbd.jit_source(model, torch.float32, "dynamics_tracks_pairs_sim") and rbd.jit_source(model, torch.float64, "dynamics_tracks_pairs_sim") is None
    # (MK_N1 / MK_NF, MK1 / MKF: the joints as the integrator stage folded into the launch sees them, csrc/rbd_mk_fuse.hpp: Atlas has 30 revolute joints and the floating base)
    assert (n1, nf) == (30, 1) and int(re.search(r"const int32_t MK1\[(\d+)\]", src).group(1)) == 90 and "rbd::MkStage F" in src
    assert (nq, nv) == (model.nq, model.nv) and 1 <= G <= 4 and 1 <= ns <= 11
    assert int(re.search(r"const int32_t RI\[(\d+)\]", src).group(1)) == 4 * ns * G
    assert int(re.search(r"const double RR\[(\d+)\]", src).group(1)) == 24 * ns * G
    assert int(re.search(r"const int32_t WK\[(\d+)\]", src).group(1)) == ns * G
    assert "M.reroot.nchain" in src  # Atlas is walked from its centre (rbd_reroot.hpp)
    lds = int(re.search(r"unsigned char lds\[(\d+)\]", src).group(1))
    assert lds % (65 * 8) == 0 and nq + 2 * nv < lds // (65 * 8) and lds <= 160 * 1024
    # every body of the tree sits in exactly one (step, track) record flagged valid (TF_VALID = bit 0 of the flags byte)
    ri = [int(x) for x in re.search(r"RI\[\d+\][^=]*= \{([^}]*)\}", src).group(1).split(",")]
    valid = sum(1 for k in range(ns * G) if (ri[4 * k + 1] >> 16) & 1)
    assert valid == model.n_bodies


def test_banked_program_of_a_mechanism(rbd):
    """The two-bodies-per-lane kernels compiled per mechanism (family 8): the level structure as compile-time constants — Atlas: 11 levels, bank 1 from level 5,
    three children to gather below the pelvis and below the upper torso and one everywhere else — and the three entry points."""
    model = rbd.load_flat_model(os.path.join(ROOT, "tests", "golden", "models", "atlas_floating.json"))
    src = rbd.jit_source(model, torch.float64, "banked")
    assert src is not None and '#include "rbd_bank.hpp"' in src
    plan = rbd.bank_plan(model)
    assert int(re.search(r"#define RBD_BANK_FIXED_NL (\d+)", src).group(1)) == int(model.levels().max()) + 1 == 11
    assert int(re.search(r"#define RBD_BANK_FIXED_L0 (\d+)", src).group(1)) == plan["L0"] == 5
    ns = [int(x) for x in re.search(r"#define RBD_BANK_FIXED_NS ([\d,]+)", src).group(1).split(",")]
    lev = model.levels()
    nch = np.zeros(model.n_bodies, int)
    for b in range(model.n_bodies):
        if model.parent[b] >= 0:
            nch[model.parent[b]] += 1
    assert ns == [0] + [int(nch[lev == l - 1].max()) for l in range(1, 11)] == [0, 3, 1, 1, 3, 1, 1, 1, 1, 1, 1]
    perm = int(re.search(r"#define RBD_BANK_FIXED_PERM (0x[0-9a-f]+)ull", src).group(1), 16)
    assert all(((perm >> l) & 1) == (ns[l] > 1) for l in range(1, 11) if l != plan["L0"])  # a hop needs the exchange column where some body has later children
    for k in ("aba_bank_spec_f64", "aba_bank_fused_spec_f64", "rnea_bank_spec_f64"):
        assert k + "(" in src
    assert "aba_bank_body<double, false, true>" in src  # Atlas: every tree joint revolute below the floating base -> the SIMPLE instantiation
    # a mechanism outside the banked scope has no such program
    assert rbd.jit_source(rbd.flatten(rbd.builders.four_bar_linkage()), torch.float64, "banked") is None or rbd.bank_plan(rbd.flatten(rbd.builders.four_bar_linkage())) is not None


def test_walk_programs_compile_without_a_device(rbd):
    """... and compile for gfx950 here (into the library's cache: the GPU tests of the same trees load them)."""
    for model in walk_trees(rbd):
        ok, log = rbd.jit_precompile(model, torch.float64)
        if ok is None:
            pytest.skip("libhiprtc not available")
        assert ok, log
        assert rbd.jit_source(model, torch.float64, "dynamics_tracks") is not None
        src = rbd.jit_source(model, torch.float64, "inverse_dynamics_tracks")
        assert src is not None and "rnea_walk_spec_f64" in src and "M.reroot.nchain" not in src  # (inverse dynamics walks the original tree)


def test_code_objects_are_the_kind_the_descriptor_rewrite_knows(rbd, tmp_path, monkeypatch):
    """The walk programs address accumulation registers by number; the library gives the wavefront all 256 of them by rewriting GRANULATED_WORKITEM_VGPR_COUNT in
    the kernel descriptor (csrc/rbd_jit.hip: jit_kd_cover_agprs), which ties it to the descriptor layout of AMDGPU HSA code objects v5 / v6.  What this build of
    hiprtc produces is checked here, field by field, the way the library checks it before it touches an object: ELF64 little-endian, OS ABI 64 (AMDGPU HSA), ABI
    version 3 or 4, machine 224, gfx950; one 64-byte `<kernel>.kd` symbol per kernel whose rsrc3 ACCUM_OFFSET is the next multiple of 4 above the kernel's VGPRs
    and whose rsrc1 granule covers them.  An object of another kind makes the library step aside to the interpreting kernel, never run with a wrong descriptor."""
    import struct
    monkeypatch.setenv("RBD_JIT_CACHE", str(tmp_path))
    # (a mechanism nobody else compiles: programs this process has compiled before are remembered in memory and would not be written to the new directory)
    model = rbd.flatten(rbd.builders.tree_mechanism(np.random.default_rng(987654), [("Revolute", [("Revolute", [("Prismatic", [])])])]))
    ok, log = rbd.jit_precompile(model, torch.float64)
    if ok is None:
        pytest.skip("libhiprtc not available")
    assert ok and "not used" not in log, log  # every walk program passed the library's own check and rewrite
    seen = 0
    for f in sorted(os.listdir(tmp_path)):
        c = open(tmp_path / f, "rb").read()
        assert c[:4] == b"\x7fELF" and c[4] == 2 and c[5] == 1 and c[7] == 64 and c[8] in (3, 4)
        assert struct.unpack_from("<H", c, 0x12)[0] == 224 and struct.unpack_from("<I", c, 0x30)[0] & 0xff == 0x4f
        shoff, = struct.unpack_from("<Q", c, 0x28)
        shentsize, shnum = struct.unpack_from("<HH", c, 0x3a)
        sh = lambda i: struct.unpack_from("<IIQQQQIIQQ", c, shoff + i * shentsize)  # name type flags addr offset size link info align entsize
        for i in range(shnum):
            _, typ, _, _, off, size, link, _, _, entsize = sh(i)
            if typ != 2:
                continue
            stroff = sh(link)[4]
            for e in range(off, off + size, entsize):
                name, info, other, shndx, value, symsize = struct.unpack_from("<IBBHQQ", c, e)
                nm = c[stroff + name:c.index(b"\0", stroff + name)].decode()
                if not nm.endswith(".kd"):
                    continue
                assert symsize == 64
                _, _, _, saddr, soff, _, _, _, _, _ = sh(shndx)
                kd = soff + value - saddr
                rsrc3, rsrc1 = struct.unpack_from("<II", c, kd + 44)
                accum_offset, granule = ((rsrc3 & 0x3f) + 1) * 4, ((rsrc1 & 0x3f) + 1) * 8
                assert 4 <= accum_offset <= 256 and granule >= accum_offset  # (before the rewrite: the VGPRs alone, or VGPRs + the allocator's own AGPRs)
                seen += 1
    assert seen >= 4


def test_a_looped_walk_program_whose_allocator_reaches_the_stash_is_refused(rbd, tmp_path):
    """The `simulate` program with four stages per launch (csrc/rbd_walk.hpp, LOOP = true) is the one walk program whose register allocator may take accumulation
    registers of its own, beside the stash the passes address by number from a255 down.  Admission is static (csrc/rbd_jit.hip jit_walk_admit, round 6; round 5 ran
    the kernel against the single-stage one inside the first rbd_simulate call): loaded only if .agpr_count <= the stash's lowest register, which the program's marker
    names.  Fed here: a kernel that deliberately takes a100 (.agpr_count = 101) — refused against a stash that reaches down to a58 (Atlas fp64: 11 steps x 9 values x 2),
    admitted against one that ends at a120, refused as an ordinary walk program (those must take none), and refused when the marker is missing the register."""
    import ctypes
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    src = tmp_path / "k.hip"
    src.write_text('#include <hip/hip_runtime.h>\nextern "C" __global__ __launch_bounds__(64) void clobber(float* x) {\n  float v = x[threadIdx.x];\n'
                   '  asm volatile("v_accvgpr_write_b32 a[100], %0" ::"v"(v) : "a100");\n  asm volatile("v_accvgpr_read_b32 %0, a[100]" : "=v"(v));\n  x[threadIdx.x] = v;\n}\n')
    obj = tmp_path / "k.hsaco"
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "--offload-device-only", "--no-gpu-bundle-output", "-O3", "-c", str(src), "-o", str(obj)],
                          stderr=subprocess.DEVNULL)
    code = obj.read_bytes()
    L = rbd._capi.lib()

    def admitted(marker):
        log = ctypes.create_string_buffer(512)
        st = L.rbd_jit_check_walk_object(marker.encode(), code, len(code), log, len(log))
        assert st in (0, 3), st
        return st == 0, log.value.decode()
    ok, log = admitted("// rbd-walk-sim-loop stash-from=a58\n")
    assert not ok and "101 accumulation registers" in log, log
    ok, log = admitted("// rbd-walk-sim-loop stash-from=a120\n")
    assert ok, log
    ok, log = admitted("// rbd-walk-sim-loop stash-from=a100\n")  # a0 .. a100 against a stash from a100: one register shared
    assert not ok
    ok, log = admitted("// rbd-walk-sim-loop stash-from=a101\n")
    assert ok, log
    assert not admitted("// a walk program of the straight-line kind\n")[0]
    assert not admitted("// rbd-walk-sim-loop\n")[0]
    # every looped program of the mechanisms in the cache names its stash: Atlas fp64 re-rooted = 256 - 2 * 9 * steps
    model = rbd.load_flat_model(os.path.join(ROOT, "tests", "golden", "models", "atlas_floating.json"))
    for dt, fam, regs in ((torch.float64, "dynamics_tracks_sim", 2), (torch.float32, "dynamics_tracks_sim", 1), (torch.float32, "dynamics_tracks_pairs_sim", 2)):
        s = rbd.jit_source(model, dt, fam)
        assert s is not None
        m = re.search(r"// rbd-walk-sim-loop stash-from=a(\d+)\n", s)
        ns = int(re.search(r"static constexpr int NS = (\d+)", s).group(1))
        assert m and int(m.group(1)) == 256 - regs * 9 * ns
