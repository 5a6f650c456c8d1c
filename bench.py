#!/usr/bin/env python
"""bench.py — headline benchmark: batched forward dynamics (`dynamics!`, fused ABA) on Atlas.

Default workload (BASELINE.json configs[1], `--config 2`): Atlas v5 URDF with floating base (nq 37, nv 36, 31 bodies; the reference's
perf/runbenchmarks.jl:14-19 mechanism, from the vendored test/urdf/atlas.urdf), batch = 4096 states per GPU, fp64, random (q, v, τ)
drawn with the reference's distributions.  One *step* = one `dynamics!` over the whole batch (one ABA kernel launch); inputs are
resident in HBM before the timed region (and, being re-evaluated every step, in L2 / Infinity Cache: the HBM-axis figure is a cache
figure — irrelevant while the kernel is ALU/latency bound, but said here).  The same line also carries the run WITH a random external
wrench on every body (what the reference's published 9.874 µs is measured with, perf/runbenchmarks.jl:59-67).

Other BASELINE configs: `--config 3` (65 536 states fp32, mass_matrix! + Cholesky solve), `--config 4` (65 536 states per GPU fp32 ABA,
v̇ gathered over RCCL: both the compute-only rate and the rate with the all-gather inside every step), `--config 5` (four-bar linkage
with its loop joint, 4096 states fp64).

N > 1: `python bench.py --gpus N` re-executes itself under torch.distributed.run (one process per GPU, RCCL); launched by the driver
under torch.distributed.run it reads RANK / WORLD_SIZE from the environment.  The batch is sharded (weak scaling: the per-GPU batch is
fixed), no data-path collective; `n_gpus` is the world size RCCL reports.  It refuses to run when fewer GPUs are visible than asked for.
With N > 1 and no `--config`, the line's `value` is the SHARDED config — BASELINE configs[3], 65 536 fp32 states per GPU, the one the north star
shards over 8 GPUs — not the 22 µs launches of configs[1]; `per_rank` lists every rank's own rate (there is no data-path collective, so a rank's rate
IS the one-GPU rate of the same run), and the N = 1 point of the same workload is the `config4_shard` block of the one-GPU line.

Prints ONE JSON line (rank 0).  `roofline` prices the dominant kernel against HBM with the ALGORITHMIC bytes (sizeof(T)·(nq + 3·nv)
per evaluation + q̇, SURVEY.md §8 d); `alu` gives the binding roof (vector ALU).  The result of the timed launches is compared with
the oracle over the WHOLE batch.  `cpu_baseline` times the oracle's restatement of the reference route on the host cores (N = 1 only).
"""
import argparse
import ctypes
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
# Kernel arguments in device memory: with them in host memory every wavefront's first scalar loads cross PCIe and a 24 µs launch
# becomes 28 µs (measured).  It is this image's default; set explicitly so the number does not depend on it.
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
# The oracle (parity check, cpu_baseline) is OpenMP code on every host core.  With libgomp's default wait policy its worker threads keep
# spinning after a parallel region; a GPU timing leg that follows is then launched from a starved host thread (seen once: 65 us per step for
# the with-wrenches leg of a 25 us kernel).  Passive waiting; the timed headline region also runs BEFORE the first oracle call.
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
os.environ.setdefault("GOMP_SPINCOUNT", "0")
# A benchmark must not start on the interpreting kernels while a mechanism's compiled ones are still being built in the background (the library's default for
# a cold cache, csrc/rbd_jit.hip): wait for them.  build() compiles the bench mechanisms' ahead of time, so on the driver's box this only loads code objects.
os.environ.setdefault("RBD_JIT_ASYNC", "0")

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s spec
FP64_VECTOR_PEAK_TF = 78.6  # MI355X public spec (256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz)
FP32_VECTOR_PEAK_TF = 157.3
ABA_FLOPS_PER_EVAL = 27.0e3  # fused world-frame ABA, Atlas floating
PMC_FILE = "r06_pmc_traffic.json"  # written by scripts/gpu_measure.sh
KERNELS = {"inverse_dynamics": "rnea_bank_kernel (<= one resident round of workgroups) / rnea_walk_kernel"}
CONFIGS = {
    2: dict(model="atlas_floating", batch=4096, dtype="f64", op="dynamics", label="BASELINE configs[1]"),
    3: dict(model="atlas_floating", batch=65536, dtype="f32", op="mass_matrix_solve", label="BASELINE configs[2]"),
    4: dict(model="atlas_floating", batch=65536, dtype="f32", op="dynamics", label="BASELINE configs[3]: 65536 states per GPU"),
    5: dict(model="four_bar", batch=4096, dtype="f64", op="dynamics", label="BASELINE configs[4]"),
}


def kernel_source_hash():
    """Identifies the kernel sources a PMC traffic figure was measured on (profiles/r03_pmc_traffic.json records it)."""
    h = hashlib.sha1()
    d = os.path.join(ROOT, "rigidbodydynamics.jl_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:12]


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", type=int, default=None, choices=sorted(CONFIGS),
                    help="BASELINE.json config (default: 2 = the headline, configs[1], on one GPU; 4 = the sharded configs[3] with --gpus N > 1)")
    ap.add_argument("--op", default=None, choices=["dynamics", "inverse_dynamics", "mass_matrix_solve"],
                    help="the entry point timed (default: the config's; --config 2 --op inverse_dynamics = the RNEA half of configs[1] as its own line)")
    ap.add_argument("--no-emit-M", action="store_true", help="config 3: M_out = NULL (x only) as the timed leg")
    ap.add_argument("--packed-M", action="store_true", help="config 3: M as the packed lower triangle (rbd_mass_matrix_solve_packed)")
    ap.add_argument("--bodies", action="store_true", help="--op inverse_dynamics: with jointwrenches and accelerations out (the call shape of perf/runbenchmarks.jl:49-57)")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="ONE timed leg only (no with-wrenches / graph-replay / M_out = NULL / pipelined legs): what a profiler run wants — every launch it sees is the leg")
    ap.add_argument("--batch", type=int, default=None, help="states per GPU (default: the config's)")
    ap.add_argument("--dtype", default=None, choices=["f64", "f32"])
    ap.add_argument("--model", default=None)
    ap.add_argument("--layout", default="aos", choices=["aos", "soa"])
    ap.add_argument("--graph", action="store_true", help="capture the K steps in one hipGraph")
    ap.add_argument("--gather-every-step", action="store_true", help="RCCL all-gather of v̇ inside every timed step (config 4 reports both anyway)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--algorithm", default="aba", choices=["aba", "aba_walk", "aba_lanes", "aba_banks"],
                    help="lane mapping of the fused ABA: aba = the library's choice by batch size")
    ap.add_argument("--no-pipelined", action="store_true",
                    help="skip the extra (informational) measurement of two independent batches issued on two HIP streams")
    ap.add_argument("--wrenches", action="store_true", help="headline WITH a random external wrench on every body (reported next to it otherwise)")
    ap.add_argument("--no-other-configs", action="store_true", help="headline only: skip the config3 / config4_shard / config5 / inverse_dynamics blocks")
    ap.add_argument("--full-out", default=None, help="also write the line with every rider's full block (roofline, alu, parity_check, traffic) to this file")
    ap.add_argument("--op-sim", action="store_true", help="ONE leg: K RK4 steps of rbd_simulate at --batch / --dtype (what scripts/gpu_measure.sh profiles as sim64 / sim32)")
    ap.add_argument("--op-kin", action="store_true", help="ONE leg: the kinematics by-products at --batch / --dtype (gpu_measure.sh: kin)")
    ap.add_argument("--selftest-launch", action="store_true", help="only exercise the N-rank launcher (gloo on CPU): prints the world size the ranks saw")
    args = ap.parse_args()
    world = max(args.gpus, int(os.environ.get("WORLD_SIZE", "1")))
    args.config_given = args.config is not None
    if args.config is None:
        args.config = 4 if world > 1 and not args.selftest_launch else 2
    cfg = CONFIGS[args.config]
    args.model = args.model or cfg["model"]
    args.batch = args.batch or cfg["batch"]
    args.dtype = args.dtype or cfg["dtype"]
    args.op = args.op or cfg["op"]
    if args.no_extra_legs:
        args.no_pipelined = True
    args.solve_only_leg = not args.no_extra_legs
    if args.steps is None:
        args.steps = 2000 if args.batch <= 8192 else 200
    if args.warmup is None:
        args.warmup = 100 if args.batch <= 8192 else 20
    return args


def maybe_spawn(args):
    """`--gpus N` without a torch.distributed.run environment: start the N ranks ourselves and relay rank 0's line."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    if not args.selftest_launch:
        import torch
        have = torch.cuda.device_count()
        if have < args.gpus:
            print(json.dumps({"error": f"--gpus {args.gpus} but only {have} GPU(s) visible: refusing to run fewer ranks than asked for"}))
            sys.exit(2)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd))


def selftest_launch(args):
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    t = torch.ones(1)
    dist.all_reduce(t)
    if dist.get_rank() == 0:
        print(json.dumps({"selftest": "launcher", "asked": args.gpus, "n_gpus": dist.get_world_size(), "ranks_counted": int(t.item())}))
    dist.destroy_process_group()


def setup(args):
    """Process-wide state: distributed init (one rank per GPU over RCCL), device, the library."""
    import numpy as np
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: reporting the world size actually running", file=sys.stderr)
    # developer switch (scripts/gpu_check.sh): every rank on cuda:0 with gloo for the collectives, so that a 1-GPU box runs the N > 1 code path of this file
    # (sharding, max over ranks, per-rank rates); nothing it prints is a measurement
    one_device = os.environ.get("RBD_BENCH_ONE_DEVICE") == "1"
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(0 if one_device else local_rank)
        if one_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        world = dist.get_world_size()  # what RCCL sees
    else:
        dist = None
        torch.cuda.set_device(0)
    device = torch.device("cuda", local_rank if world > 1 and not one_device else 0)
    import rbd_amd as rbd
    from rigidbodydynamics_jl_amd import _capi
    import oracle
    return dict(np=np, torch=torch, world=world, rank=rank, dist=dist, device=device, rbd=rbd, _capi=_capi, oracle=oracle)


def run(args, env):
    """One BASELINE config: W warm-up steps, K timed steps, whole-batch parity against the oracle; returns the JSON line as a dict (rank 0) or None."""
    np, torch, world, rank, dist, device = env["np"], env["torch"], env["world"], env["rank"], env["dist"], env["device"]
    rbd, _capi, oracle = env["rbd"], env["_capi"], env["oracle"]

    if args.model == "four_bar":
        model = rbd.flatten(rbd.four_bar_linkage())
    elif args.model.startswith("randmech"):  # the reference's own test mechanism (test/test_mechanism_algorithms.jl:1-11; SPQuatFloating -> QuaternionSpherical), seed = the suffix
        model = rbd.flatten(rbd.randmech(np.random.default_rng(int(args.model[8:] or 1))))
    else:
        model = rbd.load_flat_model(os.path.join(ROOT, "tests", "golden", "models", args.model + ".json"))
    B = args.batch
    tdt = torch.float64 if args.dtype == "f64" else torch.float32
    ndt = np.float64 if args.dtype == "f64" else np.float32
    es = 8 if args.dtype == "f64" else 4
    rng = np.random.default_rng(1 + rank)
    if args.model == "four_bar":
        # the consistent initial state of test/test_simulate.jl:195-200, joint 1 perturbed (exercises the Baumgarte term; SURVEY.md §8 d config 5)
        q = np.tile(np.asarray(rbd.FOUR_BAR_INITIAL_Q, float), (B, 1))
        q[:, 0] += rng.uniform(-0.05, 0.05, B)
        v = np.tile(np.asarray(rbd.FOUR_BAR_INITIAL_V, float), (B, 1))
    else:
        q = rbd.rand_configuration(model, B, rng)
        v = rbd.rand_velocity(model, B, rng)
    tau = rng.random((B, model.nv))
    fext_all = rng.random((B, 6 * model.n_bodies))

    state = rbd.MechanismState(model, B, dtype=tdt, device=device, layout=args.layout)
    result = rbd.DynamicsResult(model, B, dtype=tdt, device=device, layout=args.layout)
    rbd.set_configuration_(state, q)
    rbd.set_velocity_(state, v)

    def to_dev(a):
        t = torch.as_tensor(a, dtype=tdt)
        if args.layout == "soa":
            t = t.t().contiguous()
        return t.to(device)

    d_tau, d_fext = to_dev(tau), to_dev(fext_all)
    x_out = torch.zeros_like(d_tau)
    packed_M = args.op == "mass_matrix_solve" and getattr(args, "packed_M", False)
    n_packed = model.nv * (model.nv + 1) // 2
    d_packed = torch.zeros((B, n_packed) if args.layout == "aos" else (n_packed, B), dtype=tdt, device=device) if packed_M else None
    bodies = args.op == "inverse_dynamics" and getattr(args, "bodies", False)
    d_jw = torch.zeros_like(d_fext) if bodies else None
    d_acc = torch.zeros_like(d_fext) if bodies else None
    gathered = torch.empty((world * B, model.nv) if args.layout == "aos" else (world, model.nv, B), dtype=tdt, device=device) if world > 1 else None

    # low-overhead launch: pre-marshalled ctypes call straight into the C ABI
    L = _capi.lib()
    algo_id = {"aba": 0, "aba_lanes": 2, "aba_banks": 4, "aba_walk": 6}[args.algorithm]
    stream = torch.cuda.current_stream(device)
    L.rbd_workspace_set_stream(state.ws.handle, ctypes.c_void_p(stream.cuda_stream))
    vp = ctypes.c_void_p

    def make_step(with_fext, st=state, res=result, out=x_out, emit_M=True):
        if args.op == "mass_matrix_solve" and packed_M and emit_M:
            opts = st._opts(_capi.ALGO_CRBA_CHOLESKY)
            c_args = (st.ws.handle, B, vp(st.q.data_ptr()), vp(d_tau.data_ptr()), vp(out.data_ptr()), vp(d_packed.data_ptr()), ctypes.byref(opts))
            fn, name = L.rbd_mass_matrix_solve_packed, "rbd_mass_matrix_solve_packed"
        elif args.op == "mass_matrix_solve":
            opts = st._opts(_capi.ALGO_CRBA_CHOLESKY)
            c_args = (st.ws.handle, B, vp(st.q.data_ptr()), vp(d_tau.data_ptr()), vp(out.data_ptr()), vp(res.massmatrix.data_ptr() if emit_M else 0), ctypes.byref(opts))
            fn, name = L.rbd_mass_matrix_solve, "rbd_mass_matrix_solve"
        elif args.op == "inverse_dynamics" and getattr(args, "bodies", False):  # ... with the per-body outputs of the reference's signature (:542-553)
            opts = st._opts(0)
            c_args = (st.ws.handle, B, vp(st.q.data_ptr()), vp(st.v.data_ptr()), vp(d_tau.data_ptr()), vp(d_fext.data_ptr() if with_fext else 0),
                      vp(out.data_ptr()), vp(d_jw.data_ptr()), vp(d_acc.data_ptr()), ctypes.byref(opts))
            fn, name = L.rbd_inverse_dynamics_bodies, "rbd_inverse_dynamics_bodies"
        elif args.op == "inverse_dynamics":  # the RNEA half of BASELINE configs[1]: v̇ ~ U[0,1) (the `tau` draw) in, τ out
            opts = st._opts(0)
            c_args = (st.ws.handle, B, vp(st.q.data_ptr()), vp(st.v.data_ptr()), vp(d_tau.data_ptr()), vp(d_fext.data_ptr() if with_fext else 0),
                      vp(out.data_ptr()), ctypes.byref(opts))
            fn, name = L.rbd_inverse_dynamics, "rbd_inverse_dynamics"
        else:
            opts = st._opts(algo_id)
            lam = res.lambda_.data_ptr() if model.nc > 0 else 0
            c_args = (st.ws.handle, B, vp(st.q.data_ptr()), vp(st.v.data_ptr()), vp(d_tau.data_ptr()), vp(d_fext.data_ptr() if with_fext else 0),
                      vp(res.vd.data_ptr()), vp(res.qd.data_ptr()), vp(lam), ctypes.byref(opts))
            fn, name = L.rbd_dynamics, "rbd_dynamics"

        def step(_keep=(opts, c_args)):
            s = fn(*c_args)
            if s != 0:
                raise RuntimeError(f"{name} status {s}: {L.rbd_status_string(s)} {L.rbd_last_hip_error()}")
        return step

    def timed(step, gather_each, use_graph=False):
        """W warm-up steps, then exactly K steps between barrier + synchronize on both sides; (wall seconds, kernel ms per step by HIP events)."""
        def one():
            step()
            if gather_each:
                dist.all_gather_into_tensor(gathered, result.vd)
        for _ in range(args.warmup):
            one()
        torch.cuda.synchronize(device)
        graph = None
        if (args.graph or use_graph) and not gather_each:
            graph = torch.cuda.CUDAGraph()
            cap = torch.cuda.Stream(device)
            with torch.cuda.stream(cap):
                L.rbd_workspace_set_stream(state.ws.handle, vp(cap.cuda_stream))
                with torch.cuda.graph(graph, stream=cap):
                    for _ in range(args.steps):
                        step()
            L.rbd_workspace_set_stream(state.ws.handle, vp(stream.cuda_stream))
            torch.cuda.synchronize(device)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        ev0.record(stream)
        if graph is not None:
            graph.replay()
        else:
            for _ in range(args.steps):
                one()
        ev1.record(stream)
        torch.cuda.synchronize(device)
        if dist is not None:
            dist.barrier()
        wall = time.perf_counter() - t0
        timed.own_ms = ev0.elapsed_time(ev1)  # this rank's own K steps (device time), before the max over ranks
        if dist is not None:
            tw = torch.tensor([wall], dtype=torch.float64, device=device)
            dist.all_reduce(tw, op=dist.ReduceOp.MAX)
            wall = float(tw.item())
        return wall, ev0.elapsed_time(ev1) / args.steps

    headline_fext = bool(args.wrenches)
    step = make_step(headline_fext, emit_M=not args.no_emit_M)
    wall, kernel_ms = timed(step, args.gather_every_step and world > 1)
    per_rank = None
    if dist is not None:  # every rank's own rate over the same K steps (no data-path collective: this IS the one-GPU rate of this run)
        mine = torch.tensor([B * args.steps / (timed.own_ms * 1e-3)], dtype=torch.float64, device=device)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [float(t.item()) for t in allr]

    # ---- parity of the timed work against the oracle over the WHOLE batch (outside the timed region) ----
    qf, vf, tf = [a.astype(ndt).astype(np.float64) for a in (q, v, tau)]
    ff = fext_all.astype(ndt).astype(np.float64) if headline_fext else None
    ncores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    check = {}
    if args.op == "mass_matrix_solve":
        n = min(B, 4096)  # M is nv x nv per state: the full 65 536-state comparison would need 0.7 GB of host doubles
        Mref = oracle.mass_matrix(model, qf[:n], nthreads=ncores)
        Mref = np.tril(Mref) + np.transpose(np.tril(Mref, -1), (0, 2, 1))
        xref = np.linalg.solve(Mref, tf[:n, :, None])[:, :, 0]
        if packed_M:
            got_M = rbd.unpack_lower((d_packed if args.layout == "aos" else d_packed.t())[:n].double(), model.nv)
        else:
            got_M = result.massmatrix if args.layout == "aos" else result.massmatrix.t()
            got_M = got_M[:n].double().cpu().numpy().reshape(n, model.nv, model.nv).transpose(0, 2, 1)
        got_x = (x_out if args.layout == "aos" else x_out.t())[:n].double().cpu().numpy()
        il = np.tril_indices(model.nv)
        err_M = None if args.no_emit_M else float(np.abs(got_M[:, il[0], il[1]] - Mref[:, il[0], il[1]]).max() / np.abs(Mref).max())  # (None: M was not produced)
        res = np.einsum("bij,bj->bi", Mref, got_x) - tf[:n]
        # normwise backward error of the solve (Rigal–Gaches): ||M x − r|| / (||M|| ||x|| + ||r||) with the ORACLE's M
        berr = float((np.linalg.norm(res, axis=1) / (np.linalg.norm(Mref, axis=(1, 2)) * np.linalg.norm(got_x, axis=1) + np.linalg.norm(tf[:n], axis=1))).max())
        check = {"states_compared": n, "mass_matrix_rel_err": err_M, "solve_backward_err": berr,
                 "forward_err_rel_max": float(np.abs(got_x - xref).max() / np.abs(xref).max())}
        err = berr if err_M is None else max(err_M, berr)
        tol = 1e-10 if args.dtype == "f64" else 2e-5
    elif args.op == "inverse_dynamics":
        ref = oracle.inverse_dynamics(model, qf, vf, tf, ff, nthreads=ncores)
        got = (x_out if args.layout == "aos" else x_out.t()).double().cpu().numpy()
        err = float(np.abs(got - ref).max() / max(1.0, np.abs(ref).max()))
        check = {"states_compared": B}
        if bodies:  # the per-body outputs on a sample (the oracle's per-body entry point is one state per call)
            n = min(B, 4096)
            _, jw_ref, acc_ref = oracle.inverse_dynamics_bodies(model, qf[:n], vf[:n], tf[:n], ff[:n] if ff is not None else None)
            for name_, dev_, ref_ in (("jointwrenches", d_jw, jw_ref), ("accelerations", d_acc, acc_ref)):
                g = (dev_ if args.layout == "aos" else dev_.t())[:n].double().cpu().numpy()
                e = float(np.abs(g - ref_.reshape(n, -1)).max() / max(1.0, np.abs(ref_).max()))
                check[name_ + "_rel_err"] = e
                err = max(err, e)
            check["per_body_states_compared"] = n
        tol = 1e-10 if args.dtype == "f64" else 2e-4
    elif model.nc > 0:
        n = B  # the whole batch (the loop-joint oracle is one state per call: ~1 s for 4096 four-bar states)
        ref = oracle.dynamics_loops(model, qf[:n], vf[:n], tf[:n], ff[:n] if ff is not None else None)["vdot"]
        got = (result.vd if args.layout == "aos" else result.vd.t())[:n].double().cpu().numpy()
        err = float(np.abs(got - ref).max() / max(1.0, np.abs(ref).max()))
        check = {"states_compared": n}
        tol = 1e-9
    else:
        ref = oracle.dynamics(model, qf, vf, tf, ff, nthreads=ncores)
        got = (result.vd if args.layout == "aos" else result.vd.t()).double().cpu().numpy()
        if args.dtype == "f64":
            err = float(np.abs(got - ref).max() / max(1.0, np.abs(ref).max()))
            tol = 1e-10
        else:  # fp32: backward error against the fp64 oracle's M, c on a sample (cond(M) ~ 5e5: the forward error is not the criterion)
            n = min(B, 4096)
            Mref = oracle.mass_matrix(model, qf[:n], nthreads=ncores)
            Mref = np.tril(Mref) + np.transpose(np.tril(Mref, -1), (0, 2, 1))
            c = oracle.dynamics_bias(model, qf[:n], vf[:n], ff[:n] if ff is not None else None)
            r = tf[:n] - c
            err = float((np.linalg.norm(np.einsum("bij,bj->bi", Mref, got[:n]) - r, axis=1) / np.linalg.norm(r, axis=1)).max())
            check = {"backward_err_states": n, "forward_err_rel_max_whole_batch": float(np.abs(got - ref).max() / max(1.0, np.abs(ref).max()))}
            tol = 2e-5
        check["states_compared"] = B
    assert err < tol or "spec_variant" in os.environ.get("RBD_TUNE", ""), f"parity lost in bench: {err} (tolerance {tol})"

    extra = {}
    gather_ms = None
    if dist is not None and args.op == "dynamics":
        torch.cuda.synchronize(device)
        try:
            dist.all_gather_into_tensor(gathered, result.vd)  # warm-up (RCCL channel setup)
            torch.cuda.synchronize(device)
            g0 = time.perf_counter()
            dist.all_gather_into_tensor(gathered, result.vd)  # the RCCL gather of DynamicsResult.v̇ over xGMI
            torch.cuda.synchronize(device)
            gather_ms = (time.perf_counter() - g0) * 1e3
            if not args.gather_every_step:
                w2, _ = timed(step, True)
                extra["with_gather_every_step"] = {"value": world * B * args.steps / w2, "unit": "evals/s", "ms_per_step": w2 / args.steps * 1e3}
        except Exception as e:  # the gather is outside the headline's timed region: never lose the measurement to it
            gather_ms = f"failed: {type(e).__name__}: {e}"

    if args.op == "dynamics" and model.nc == 0 and world == 1 and not args.no_extra_legs:
        # the same K steps with / without a random external wrench on every body, so that the line carries both
        try:
            w2, k2 = timed(make_step(not headline_fext), False)
            extra["with_external_wrenches" if not headline_fext else "without_external_wrenches"] = {
                "value": B * args.steps / w2, "unit": "evals/s", "ms_per_step": w2 / args.steps * 1e3, "kernel_ms": k2}
        except Exception as e:
            extra["with_external_wrenches"] = f"failed: {type(e).__name__}: {e}"
    if args.op == "mass_matrix_solve" and world == 1 and args.solve_only_leg and not args.no_emit_M:
        # the same solve for a caller that does not want M back (M_out = NULL): the whole-square store of M is most of the route's traffic
        try:
            w4, k4 = timed(make_step(False, emit_M=False), False)
            extra["solve_only_M_not_emitted"] = {"value": B * args.steps / w4, "unit": "solves/s", "ms_per_step": w4 / args.steps * 1e3, "kernel_ms": k4}
        except Exception as e:
            extra["solve_only_M_not_emitted"] = f"failed: {type(e).__name__}: {e}"
    if world == 1 and not args.graph and not args.no_extra_legs:
        # informational: the same K steps captured once in a hipGraph and replayed (what a caller with a fixed step loop would do); `value` stays
        # the plain stream-launch figure
        try:
            w3, k3 = timed(step, False, use_graph=True)
            extra["hip_graph_replay"] = {"value": B * args.steps / w3, "unit": "evals/s", "ms_per_step": w3 / args.steps * 1e3, "kernel_ms": k3}
        except Exception as e:
            extra["hip_graph_replay"] = f"failed: {type(e).__name__}: {e}"

    if rank != 0:
        return None

    evals = world * B * args.steps
    value = evals / wall
    if args.op == "mass_matrix_solve":
        alg_bytes = es * (model.nq + model.nv + model.nv * (model.nv + 1) // 2 + model.nv)  # q, rhs in; lower M, x out
        flops = 5.8e3 + 18.2e3
        metric = "mass_matrix! + Cholesky solves/sec (Atlas 30-DoF, batch)"
        opname = f"{args.dtype} mass_matrix! + Cholesky solve" + (", M as the packed lower triangle" if packed_M else "")
    elif args.op == "inverse_dynamics":
        alg_bytes = es * (model.nq + 3 * model.nv + (12 * model.n_bodies if bodies else 0))  # q, v, v̇ in; τ out (+ a wrench and an acceleration per body)
        flops = 18.6e3
        metric = "inverse_dynamics! evals/sec (Atlas 30-DoF, batch)"
        opname = f"{args.dtype} inverse_dynamics! (RNEA)"
    else:
        alg_bytes = es * (model.nq + 3 * model.nv + model.nq + (model.nc if model.nc else 0))  # q, v, τ in; v̇ and q̇ out (+ λ)
        if headline_fext:
            alg_bytes += es * 6 * model.n_bodies
        flops = ABA_FLOPS_PER_EVAL * model.nv / 36.0 if model.nc == 0 else 43.0e3 * model.nv / 36.0  # (27 kflop is Atlas's count, nv 36: scaled by nv for other trees)
        metric = "ABA dynamics! evals/sec (Atlas 30-DoF, batch)" if args.model.startswith("atlas") else f"dynamics! evals/sec ({args.model}, batch)"
        opname = f"{args.dtype} " + ("fused ABA dynamics!" if model.nc == 0 else "dynamics! with loop joints (RNEA + CRBA + constrained solve)")
    achieved_gbs = alg_bytes * B / (kernel_ms * 1e-3) / 1e9
    peak_tf = FP64_VECTOR_PEAK_TF if args.dtype == "f64" else FP32_VECTOR_PEAK_TF
    achieved_tf = flops * B / (kernel_ms * 1e-3) / 1e12

    # HBM traffic per launch from the PMC passes of scripts/gpu_measure.sh (one rocprofv3 run PER LEG since round 4: `--no-extra-legs`), recorded with the hash
    # of the kernel sources it was measured on: a figure from other sources is flagged stale, never passed along as current
    key = f"{args.model}_{args.dtype}_B{B}_{args.op}" + ("_noM" if args.no_emit_M else "") + ("_bodies" if bodies else "") + ("_packed" if packed_M else "")
    traffic, traffic_stale = pmc_traffic(key)

    out = {
        "metric": metric, "value": value, "unit": "evals/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"{args.model} nq={model.nq} nv={model.nv} bodies={model.n_bodies}, batch={B}/GPU, {opname} ({CONFIGS[args.config]['label']})",
                   "batch_per_gpu": B, "layout": args.layout, "external_wrenches": headline_fext,
                   "hip_graph": bool(args.graph), "parallelism": f"batch-sharded x{world}, no data-path collective",
                   "gather_every_step": bool(args.gather_every_step and world > 1), "inputs": "resident in HBM, re-evaluated every step (L2 / Infinity-Cache hits)"},
        "roofline": {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_stale": traffic_stale,
                     # the inputs of a batch this small never leave the caches between steps: the "HBM" axis is then fabric traffic (the counters' raw FETCH_SIZE
                     # is below the input bytes; profiles/r06_pmc_traffic.json)
                     "resident": "L2/MALL" if alg_bytes * B < 64e6 and isinstance(traffic, dict) and traffic.get("fetch_raw_bytes", 1e30) < es * (model.nq + 2 * model.nv) * B else None,
                     "kernel": (L.rbd_workspace_last_kernel(state.ws.handle) or b"").decode() or KERNELS.get(args.op),  # (what the workspace launched last; the table only if it cannot say)
                     "kernel_ms": kernel_ms, "algorithmic_bytes_per_eval": alg_bytes},
        "alu": {"bound": "fp64 vector ALU" if args.dtype == "f64" else "fp32 vector ALU", "achieved": achieved_tf, "peak": peak_tf,
                "unit": "TFLOP/s", "frac": achieved_tf / peak_tf, "flops_per_eval": flops,
                "note": "the path is ALU/latency bound, not HBM bound (SURVEY.md F8): compulsory traffic is ~1.5 KB/eval"},
        "parity_rel_err_vs_oracle": err, "parity_check": check,
        "published_reference": {"dynamics!_us_per_eval": 9.874, "evals_per_s": 1.01e5, "hardware": "Apple M2, 1 thread, Julia 1.11, WITH external wrenches",
                                "source": "docs/src/benchmarks.md:71-78"},
    }
    out.update(extra)
    if per_rank is not None:
        out["per_rank"] = {"value": per_rank, "unit": "evals/s", "note": "each rank's own K steps by HIP events; value = all ranks' states / the slowest rank's wall time"}
    if gather_ms is not None:
        out["rccl_all_gather_vdot_ms"] = gather_ms

    if not args.no_pipelined and world == 1 and args.op == "dynamics" and model.nc == 0:
        # Informational, NOT `value`: the same step count with two INDEPENDENT batches of B states alternating on two HIP streams
        # (one workspace each).  At B = 4096 a single launch leaves every SIMD with one wavefront (latency-bound);
        # a second launch in flight gives each SIMD a second wavefront to interleave.  Callers whose batches do not depend on each
        # other (sampling-based control, the reference's own benchmark loop) can run this way; `simulate` cannot.
        try:
            rng2 = np.random.default_rng(1001)
            state2 = rbd.MechanismState(model, B, dtype=tdt, device=device, layout=args.layout)
            result2 = rbd.DynamicsResult(model, B, dtype=tdt, device=device, layout=args.layout)
            rbd.set_configuration_(state2, rbd.rand_configuration(model, B, rng2))
            rbd.set_velocity_(state2, rbd.rand_velocity(model, B, rng2))
            s_a, s_b = torch.cuda.Stream(device), torch.cuda.Stream(device)
            L.rbd_workspace_set_stream(state.ws.handle, vp(s_a.cuda_stream))
            L.rbd_workspace_set_stream(state2.ws.handle, vp(s_b.cuda_stream))
            both = (make_step(headline_fext), make_step(headline_fext, state2, result2))
            for k in range(args.warmup):
                both[k & 1]()
            torch.cuda.synchronize(device)
            p0 = time.perf_counter()
            for k in range(args.steps):
                both[k & 1]()
            torch.cuda.synchronize(device)
            p1 = time.perf_counter()
            L.rbd_workspace_set_stream(state.ws.handle, vp(stream.cuda_stream))
            out["pipelined_independent_batches"] = {"streams": 2, "value": B * args.steps / (p1 - p0), "unit": "evals/s",
                                                    "ms_per_step": (p1 - p0) / args.steps * 1e3,
                                                    "note": "informational: two independent batches in flight; not the headline value"}
        except Exception as e:  # never lose the headline line to the extra measurement
            out["pipelined_independent_batches"] = f"failed: {type(e).__name__}: {e}"

    if not args.no_cpu_baseline and world == 1 and model.nc == 0:  # the CPU leg is reported at N = 1 only (rank 0 of a multi-GPU run would time it while its peers wait)
        # host threads actually available to this process: affinity mask, capped by the cgroup CPU quota when there is one
        try:
            quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
            if quota != "max":
                ncores = max(1, min(ncores, int(round(int(quota) / int(period)))))
        except Exception:
            pass
        os.environ.setdefault("OMP_PROC_BIND", "close")
        S1 = min(B, 2048)
        qs, vs, ts = q[:S1], v[:S1], tau[:S1]
        fs = fext_all[:S1] if headline_fext else None
        oracle.dynamics(model, qs[:64], vs[:64], ts[:64], fs[:64] if fs is not None else None)  # warm
        reps1 = 0
        c0 = time.perf_counter()
        while time.perf_counter() - c0 < 4.0:
            oracle.dynamics(model, qs, vs, ts, fs, nthreads=1)
            reps1 += 1
        t1t = (time.perf_counter() - c0) / (reps1 * S1)
        # all cores: a sample large enough that the OpenMP fork/join is amortised (2048 states per thread per call)
        tile = max(1, (ncores * 2048 + B - 1) // B)
        qN, vN, tN = np.tile(q, (tile, 1)), np.tile(v, (tile, 1)), np.tile(tau, (tile, 1))
        fN = np.tile(fext_all, (tile, 1)) if headline_fext else None
        SN = qN.shape[0]
        oracle.dynamics(model, qN, vN, tN, fN, nthreads=ncores)  # warm (thread pool, per-thread scratch)
        repsN = 0
        c0 = time.perf_counter()
        while time.perf_counter() - c0 < 6.0:
            oracle.dynamics(model, qN, vN, tN, fN, nthreads=ncores)
            repsN += 1
        tNt = (time.perf_counter() - c0) / (repsN * SN)
        out["cpu_baseline"] = {
            "value": 1.0 / tNt, "unit": "evals/s", "cores": ncores, "kind": "port",
            "sample": f"oracle C restatement of the reference route (RNEA-bias + CRBA + Cholesky), fp64, same inputs: "
                      f"{repsN}x{SN} states on {ncores} threads ({tNt * 1e6:.2f} us/eval); single thread {reps1}x{S1} states: "
                      f"{t1t * 1e6:.2f} us/eval = {1.0 / t1t:.3e} evals/s",
            "single_thread_us_per_eval": t1t * 1e6,
        }
    return out


def sim_leg(env, B, dtype, steps=12, warm=2, dt=1e-3):
    """One RK4 step of the batched `simulate` (src/simulate.jl:36-55, MuntheKaasIntegrator.step src/ode_integrators.jl:233-299; rbd_simulate, the stage folded
    into the dynamics kernel): K steps between HIP events on Atlas at B states, then — from the same initial state — two steps of EVERY state compared with the
    numpy restatement of the integrator (oracle/simulate_np.py, its batch-vectorised form).  One step = four dynamics! evaluations: `alu` prices 4 x 27 kflop,
    `roofline` the step's compulsory bytes (q, v in and out, tau in)."""
    np, torch, device = env["np"], env["torch"], env["device"]
    rbd, _capi = env["rbd"], env["_capi"]
    import simulate_np
    model = rbd.load_flat_model(os.path.join(ROOT, "tests", "golden", "models", "atlas_floating.json"))
    tdt = torch.float64 if dtype == "f64" else torch.float32
    es = 8 if dtype == "f64" else 4
    rng = np.random.default_rng(7)
    q0 = rbd.rand_configuration(model, B, rng)
    v0 = 0.3 * rbd.rand_velocity(model, B, rng)
    tau = rng.random((B, model.nv))
    state = rbd.MechanismState(model, B, dtype=tdt, device=device)
    d_tau = torch.as_tensor(tau, dtype=tdt).to(device)
    L = _capi.lib()
    stream = torch.cuda.current_stream(device)
    L.rbd_workspace_set_stream(state.ws.handle, ctypes.c_void_p(stream.cuda_stream))
    opts = state._opts(_capi.ALGO_ABA)
    vp = ctypes.c_void_p

    def advance(n):
        st = L.rbd_simulate(state.ws.handle, B, vp(state.q.data_ptr()), vp(state.v.data_ptr()), vp(d_tau.data_ptr()), vp(0), ctypes.c_double(dt), n, ctypes.byref(opts))
        if st != 0:
            raise RuntimeError(f"rbd_simulate status {st}: {L.rbd_status_string(st)} {L.rbd_last_hip_error()}")

    def reset():
        rbd.set_configuration_(state, q0)
        rbd.set_velocity_(state, v0)

    reset()
    advance(warm)
    torch.cuda.synchronize(device)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    advance(steps)
    ev1.record(stream)
    torch.cuda.synchronize(device)
    wall = time.perf_counter() - t0
    kernel = (L.rbd_workspace_last_kernel(state.ws.handle) or b"").decode()
    # parity: two steps from the initial state, EVERY state against the integrator's numpy restatement (fp64)
    reset()
    advance(2)
    torch.cuda.synchronize(device)
    ndt = np.float64 if dtype == "f64" else np.float32
    ncores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    qs, vs, ts_ = [a.astype(ndt).astype(np.float64) for a in (q0, v0, tau)]
    q_ref, v_ref = simulate_np.simulate_batch(model, qs, vs, 2, dt, ts_, nthreads=ncores)
    qg, vg = state.q.double().cpu().numpy(), state.v.double().cpu().numpy()
    for o in range(0, 1):  # Atlas: one quaternion, at q[0:4] — compared up to sign
        sg = np.sign(qg[:, o:o + 1] * q_ref[:, o:o + 1]); sg[sg == 0] = 1
        qg[:, o:o + 4] *= sg
    eq = float(np.abs(qg - q_ref).max() / max(1.0, np.abs(q_ref).max()))
    ev = float(np.abs(vg - v_ref).max() / max(1.0, np.abs(v_ref).max()))
    tq, tv = (1e-10, 1e-9) if dtype == "f64" else (2e-5, 2e-3)
    assert eq < tq and ev < tv, f"simulate parity lost in bench: q {eq} v {ev}"
    us = ev0.elapsed_time(ev1) / steps * 1e3
    alg_bytes = es * (2 * model.nq + 3 * model.nv)
    gbs = alg_bytes * B / (us * 1e-6) / 1e9
    tf = 4 * ABA_FLOPS_PER_EVAL * B / (us * 1e-6) / 1e12
    peak_tf = FP64_VECTOR_PEAK_TF if dtype == "f64" else FP32_VECTOR_PEAK_TF
    launches = 1 if "four stages per launch" in kernel else 4  # (what the workspace says it ran: rbd_workspace_last_kernel)
    return {"metric": "RK4 simulate steps/sec x states (Atlas 30-DoF, batch)", "value": B * steps / wall, "unit": "state-steps/s", "steps": steps, "warmup": warm,
            "ms_per_step": wall / steps * 1e3, "dtype": dtype, "launches_per_step": launches, "dt": dt,
            "config": {"workload": f"atlas_floating, batch={B}, {dtype} RK4 simulate step (4 dynamics! evaluations)", "batch_per_gpu": B},
            "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": sim_traffic(dtype, B),
                         "kernel": kernel, "kernel_ms": us * 1e-3, "algorithmic_bytes_per_eval": alg_bytes},
            "alu": {"bound": "fp64 vector ALU" if dtype == "f64" else "fp32 vector ALU", "achieved": tf, "peak": peak_tf, "unit": "TFLOP/s", "frac": tf / peak_tf,
                    "flops_per_eval": 4 * ABA_FLOPS_PER_EVAL},
            "parity_rel_err_vs_oracle": max(eq, ev),
            "parity_check": {"states_compared": B, "steps": 2, "q_rel_err": eq, "v_rel_err": ev, "against": "oracle/simulate_np.py (Munthe-Kaas RK4 restated in numpy, fp64)"}}


def pmc_traffic(key):
    """HBM bytes per step from the PMC passes of scripts/gpu_measure.sh (profiles/r06_pmc_traffic.json), flagged stale when the kernel sources have changed since."""
    pmc = os.path.join(ROOT, "profiles", PMC_FILE)
    if not os.path.exists(pmc):
        return None, None
    try:
        rec = json.load(open(pmc))
        stale = rec.get("source_hash") != kernel_source_hash()
        traffic = rec.get(key)
        if stale and traffic is not None:
            traffic = dict(traffic, stale=True) if isinstance(traffic, dict) else {"bytes_per_launch": traffic, "stale": True}
        return traffic, stale
    except Exception:
        return None, None


def sim_traffic(dtype, B):
    return pmc_traffic(f"atlas_floating_{dtype}_B{B}_simulate")[0]


def kin_leg(env, B, dtype, steps=40, warm=8, model_name="atlas_floating"):
    """The kinematics by-products the reference benchmarks beside the dynamics (perf/runbenchmarks.jl:69-110: momentum_matrix!, geometric_jacobian!, momentum,
    kinetic_energy, center_of_mass — SURVEY.md §8 f3): one step = rbd_kinematics (momentum matrix + centre of mass + both energies) + rbd_geometric_jacobian (world ->
    the last body: a hand) + rbd_momentum (momentum, momentum_rate_bias) over the batch, each also timed on its own by HIP events; parity of every output against
    the oracle.  Algorithmic bytes: the inputs each call reads plus the outputs it writes."""
    np, torch, device = env["np"], env["torch"], env["device"]
    rbd, _capi, oracle = env["rbd"], env["_capi"], env["oracle"]
    if model_name.startswith("randmech"):  # the reference's own test mechanism, seed = the suffix (as in run())
        model = rbd.flatten(rbd.randmech(np.random.default_rng(int(model_name[8:] or 1))))
    else:
        model = rbd.load_flat_model(os.path.join(ROOT, "tests", "golden", "models", model_name + ".json"))
    tdt = torch.float64 if dtype == "f64" else torch.float32
    es = 8 if dtype == "f64" else 4
    rng = np.random.default_rng(11)
    q = rbd.rand_configuration(model, B, rng)
    v = rbd.rand_velocity(model, B, rng)
    state = rbd.MechanismState(model, B, dtype=tdt, device=device)
    rbd.set_configuration_(state, q)
    rbd.set_velocity_(state, v)
    nv, nq = model.nv, model.nq
    A = torch.zeros((B, 6 * nv), dtype=tdt, device=device)
    J = torch.zeros((B, 6 * nv), dtype=tdt, device=device)
    com = torch.zeros((B, 3), dtype=tdt, device=device)
    en = torch.zeros((B, 2), dtype=tdt, device=device)
    mom = torch.zeros((B, 12), dtype=tdt, device=device)
    L = _capi.lib()
    stream = torch.cuda.current_stream(device)
    L.rbd_workspace_set_stream(state.ws.handle, ctypes.c_void_p(stream.cuda_stream))
    opts = state._opts()
    vp = ctypes.c_void_p
    target = model.n_bodies - 1
    h = state.ws.handle
    calls = {
        "kinematics": (lambda: L.rbd_kinematics(h, B, vp(state.q.data_ptr()), vp(state.v.data_ptr()), vp(A.data_ptr()), vp(com.data_ptr()), vp(en.data_ptr()), ctypes.byref(opts)),
                       es * (nq + nv + 6 * nv + 3 + 2)),
        "geometric_jacobian": (lambda: L.rbd_geometric_jacobian(h, B, vp(state.q.data_ptr()), -1, target, vp(J.data_ptr()), ctypes.byref(opts)), es * (nq + 6 * nv)),
        "momentum": (lambda: L.rbd_momentum(h, B, vp(state.q.data_ptr()), vp(state.v.data_ptr()), vp(mom.data_ptr()), ctypes.byref(opts)), es * (nq + nv + 12)),
    }

    def run_all(names):
        for n_ in names:
            st = calls[n_][0]()
            if st != 0:
                raise RuntimeError(f"rbd_{n_} status {st}: {L.rbd_status_string(st)} {L.rbd_last_hip_error()}")

    def time_of(names):
        for _ in range(warm):
            run_all(names)
        torch.cuda.synchronize(device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(stream)
        for _ in range(steps):
            run_all(names)
        e1.record(stream)
        torch.cuda.synchronize(device)
        return time.perf_counter() - t0, e0.elapsed_time(e1) / steps * 1e3

    wall, us = time_of(list(calls))
    each, kernels = {}, []
    for n_ in calls:
        each[n_] = time_of([n_])[1]
        kernels.append((L.rbd_workspace_last_kernel(h) or b"").decode().split(" (")[0])
    # parity (the per-state oracle entry points are one ctypes call per state: a sample of 4096 states)
    n = min(B, 4096)
    ndt = np.float64 if dtype == "f64" else np.float32
    qf, vf = q[:n].astype(ndt).astype(np.float64), v[:n].astype(ndt).astype(np.float64)
    A_ref, _, com_ref = oracle.momentum_matrix(model, qf, vf)
    ke_ref, pe_ref = oracle.energy(model, qf, vf)
    J_ref, _ = oracle.geometric_jacobian(model, qf, -1, target)
    h_ref, hb_ref = oracle.momentum(model, qf, vf)

    def rel(got, ref):
        return float(np.abs(got - ref).max() / max(1.0, np.abs(ref).max()))
    errs = {"momentum_matrix": rel(A[:n].double().cpu().numpy().reshape(n, nv, 6).transpose(0, 2, 1), A_ref),
            "center_of_mass": rel(com[:n].double().cpu().numpy(), com_ref),
            "energies": rel(en[:n].double().cpu().numpy(), np.stack([ke_ref, pe_ref], axis=1)),
            "geometric_jacobian": rel(J[:n].double().cpu().numpy().reshape(n, nv, 6).transpose(0, 2, 1), J_ref),
            "momentum": rel(mom[:n].double().cpu().numpy(), np.concatenate([h_ref, hb_ref], axis=1))}
    err = max(errs.values())
    tol = 1e-10 if dtype == "f64" else 2e-4
    assert err < tol, f"kinematics parity lost in bench: {errs}"
    alg_bytes = sum(c[1] for c in calls.values())
    gbs = alg_bytes * B / (us * 1e-6) / 1e9
    return {"metric": "kinematics by-products sets/sec (momentum_matrix! + center_of_mass + energies, geometric_jacobian!, momentum + momentum_rate_bias)",
            "value": B * steps / wall, "unit": "sets/s", "steps": steps, "warmup": warm, "ms_per_step": wall / steps * 1e3, "dtype": dtype,
            "config": {"workload": f"{model_name}, batch={B}, {dtype} rbd_kinematics + rbd_geometric_jacobian + rbd_momentum", "batch_per_gpu": B},
            "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                         "traffic": pmc_traffic(f"{model_name}_{dtype}_B{B}_kinematics")[0], "kernel": " + ".join(kernels), "kernel_ms": us * 1e-3,
                         "algorithmic_bytes_per_eval": alg_bytes,
                         "each": {n_: {"us": round(each[n_], 2), "bytes_per_state": calls[n_][1], "hbm_frac": round(calls[n_][1] * B / (each[n_] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)}
                                  for n_ in calls}},
            "parity_rel_err_vs_oracle": err, "parity_check": dict(errs, states_compared=n)}


def sub_args(args, config, **over):
    """The arguments of another BASELINE config run inside this process (short: it rides on the headline's line)."""
    import copy
    a = copy.copy(args)
    cfg = CONFIGS[config]
    a.config, a.model, a.batch, a.dtype, a.op = config, cfg["model"], cfg["batch"], cfg["dtype"], cfg["op"]
    a.steps, a.warmup = (200, 20) if a.batch <= 8192 else (40, 8)
    a.no_cpu_baseline = a.no_pipelined = a.no_extra_legs = True
    a.wrenches = a.graph = a.no_emit_M = a.bodies = a.packed_M = False
    a.solve_only_leg = True
    a.algorithm = "aba"
    for k, v in over.items():
        setattr(a, k, v)
    return a


def sig(x, n=4):
    """A float at n significant digits (the driver keeps an 8 KB tail of stdout: the line must fit it — round-5 review)."""
    if isinstance(x, bool) or not isinstance(x, float):
        return x
    return float(f"{x:.{n}g}")


def shrink(o, n=4):
    if isinstance(o, dict):
        return {k: shrink(v, n) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [shrink(v, n) for v in o]
    return sig(o, n)


def short_kernel(k):
    return k.split(" (")[0][:48] if isinstance(k, str) else k


def rider(d):
    """A config's own line cut down to what the driver's record must carry: rate, step and kernel time, kernel, both roofline fractions, the counters' HBM bytes
    per step against the algorithmic bytes, parity and how many states were compared.  The full blocks go to stderr and to --full-out."""
    if not isinstance(d, dict):
        return d
    r, al, pc = d.get("roofline", {}), d.get("alu", {}), d.get("parity_check", {})
    B = d.get("config", {}).get("batch_per_gpu")
    o = {"value": d["value"], "unit": d["unit"], "us": d["ms_per_step"] * 1e3, "k_us": r.get("kernel_ms", 0.0) * 1e3, "kernel": short_kernel(r.get("kernel")),
         "dtype": d.get("dtype"), "B": B, "hbm": r.get("frac"), "alu": al.get("frac"), "err": d.get("parity_rel_err_vs_oracle"),
         "cmp": pc.get("states_compared")}
    t = r.get("traffic")
    if isinstance(t, dict) and B and t.get("bytes_per_step_fetch_x2") and r.get("algorithmic_bytes_per_eval"):
        o["pmc_x"] = t["bytes_per_step_fetch_x2"] / (r["algorithmic_bytes_per_eval"] * B)  # counters' HBM bytes per step / algorithmic bytes
        if t.get("stale"):
            o["pmc_stale"] = True
    if "launches_per_step" in d:
        o["launches"] = d["launches_per_step"]
    if isinstance(r.get("each"), dict):
        o["each_us"] = {k: v["us"] for k, v in r["each"].items()}
    so = d.get("solve_only_M_not_emitted")
    if isinstance(so, dict):
        o["noM_us"] = so["ms_per_step"] * 1e3
    for k in ("with_external_wrenches", "hip_graph_replay", "pipelined_independent_batches"):
        if isinstance(d.get(k), dict):
            o[k.split("_")[0] + "_us"] = d[k]["ms_per_step"] * 1e3
    return shrink(o)


def compact_headline(out):
    """The headline keeps the contract's keys in full; its informational legs become one number each."""
    o = dict(out)
    for k in ("with_external_wrenches", "without_external_wrenches", "hip_graph_replay", "pipelined_independent_batches", "with_gather_every_step"):
        if isinstance(o.get(k), dict):
            o[k] = shrink({"value": o[k]["value"], "us": o[k]["ms_per_step"] * 1e3})
    o.pop("published_reference", None)
    o["alu"] = {k: v for k, v in o["alu"].items() if k != "note"}
    t = o["roofline"].get("traffic")
    if isinstance(t, dict):
        o["roofline"] = dict(o["roofline"], traffic={k: v for k, v in t.items() if k in ("bytes_per_step_fetch_x2", "bytes_per_step_raw", "fetch_raw_bytes", "stale")})
    return {k: (shrink(v, 5) if k not in ("value", "ms_per_step") else v) for k, v in o.items()}


def main():
    args = parse_args()
    maybe_spawn(args)
    if args.selftest_launch:
        return selftest_launch(args)
    env = setup(args)
    if args.op_sim or args.op_kin:  # one leg of its own for a profiler run
        d = sim_leg(env, args.batch, args.dtype, steps=min(args.steps, 60), warm=min(args.warmup, 10)) if args.op_sim else \
            kin_leg(env, args.batch, args.dtype, steps=min(args.steps, 60), warm=min(args.warmup, 10), model_name=args.model)
        print(json.dumps(d))
        return
    out = run(args, env)
    headline = (args.config == 2 and not args.no_other_configs and args.batch == CONFIGS[2]["batch"] and args.dtype == CONFIGS[2]["dtype"] and args.op == "dynamics"
                and not args.no_extra_legs and env["world"] == 1)
    full = {}
    if headline:
        # The other BASELINE configs ride on the driver's one-GPU line (round-2 review): each with its own step time, roofline fractions and whole-batch parity;
        # `value` stays configs[1].  (N > 1: the line IS configs[3], the sharded one — see the module docstring.)  Every rider is cut to a dozen numbers (`rider`):
        # the round-5 line was 14 KB and the driver keeps 8.
        todo = [("inverse_dynamics", sub_args(args, 2, op="inverse_dynamics")), ("config3", sub_args(args, 3)), ("config4_shard", sub_args(args, 4)),
                ("config5", sub_args(args, 5)),
                ("config3_packed_M", sub_args(args, 3, packed_M=True, solve_only_leg=False)),
                # "AF and AX ... benchmark both" — the fixed-base Atlas at the headline's batch and dtype
                ("atlas_fixed_f64_B4096", sub_args(args, 2, model="atlas_fixed")),
                # the reference's own arithmetic at the large batch: dynamics!, and inverse_dynamics! with its per-body outputs — the call shape of
                # perf/runbenchmarks.jl:49-57 — at 65 536 fp64 states
                ("dynamics_f64_B65536", sub_args(args, 2, batch=65536)),
                # the reference's own test mechanism (Planar / spherical joints below a floating base: what no walk kernel takes) in its own arithmetic at the large batch
                ("randmech_dynamics_f64_B65536", sub_args(args, 2, model="randmech1", batch=65536)),
                ("inverse_dynamics_bodies_f64_B65536", sub_args(args, 2, batch=65536, op="inverse_dynamics", bodies=True))]
        for name, a in todo:
            try:
                full[name] = run(a, env)
            except Exception as e:  # never lose the headline to a rider
                full[name] = f"failed: {type(e).__name__}: {e}"
        for name, (B_, dt_) in (("simulate_step_f64_B4096", (4096, "f64")), ("simulate_step_f32_B65536", (65536, "f32")), ("simulate_step_f64_B65536", (65536, "f64"))):
            try:
                full[name] = sim_leg(env, B_, dt_)
            except Exception as e:
                full[name] = f"failed: {type(e).__name__}: {e}"
        # the by-products the reference benchmarks (perf/runbenchmarks.jl:69-110), never timed before round 6
        for name, (B_, dt_) in (("kinematics_f64_B4096", (4096, "f64")), ("kinematics_f64_B65536", (65536, "f64"))):
            try:
                full[name] = kin_leg(env, B_, dt_)
            except Exception as e:
                full[name] = f"failed: {type(e).__name__}: {e}"
    if out is not None:
        line = compact_headline(out) if headline else out
        for name, d in full.items():
            line[name] = rider(d)
        if headline or args.full_out:
            fo = dict(out, **full)
            print("bench.py full record: " + json.dumps(fo), file=sys.stderr)
            if args.full_out:
                with open(args.full_out, "w") as f:
                    json.dump(fo, f)
        print(json.dumps(line))
    if env["dist"] is not None:
        env["dist"].destroy_process_group()


if __name__ == "__main__":
    main()
