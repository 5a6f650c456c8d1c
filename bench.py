#!/usr/bin/env python
"""bench.py — headline benchmark: batched forward dynamics (`dynamics!`, fused ABA) on Atlas.

Workload (BASELINE.json configs[1]): Atlas v5 URDF with floating base (nq 37, nv 36, 31 bodies; the reference's
perf/runbenchmarks.jl:14-19 mechanism, from the vendored test/urdf/atlas.urdf), batch = 4096 states per GPU, fp64,
random (q, v, τ) drawn with the reference's distributions.  One *step* = one `dynamics!` over the whole batch
(one ABA kernel launch — aba_bank_kernel at this batch); inputs are resident in HBM before the timed region.  N > 1: one process per GPU, the
batch is sharded (weak scaling: 4096 states per GPU), no data-path collective; the RCCL all-gather of v̇ is run
once after the timed region (and inside it with --gather-every-step).

Prints ONE JSON line (rank 0).  `roofline` prices the dominant kernel against HBM with the ALGORITHMIC bytes
(sizeof(T)·(nq + 3·nv) per evaluation, SURVEY.md §8 d); `alu` gives the honest binding roof (fp64 vector ALU).
`cpu_baseline` times the oracle's restatement of the reference route (CRBA + RNEA + Cholesky) on the host cores.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
# Kernel arguments in device memory: with them in host memory every wavefront's first scalar loads cross PCIe and a 24 µs launch
# becomes 28 µs (measured).  It is this image's default; set explicitly so the number does not depend on it.
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s spec
FP64_VECTOR_PEAK_TF = 78.6  # MI355X public spec (256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz)
FP32_VECTOR_PEAK_TF = 157.3
ABA_FLOPS_PER_EVAL = 27.0e3  # fused world-frame ABA, Atlas floating


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--batch", type=int, default=4096, help="states per GPU")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--model", default="atlas_floating")
    ap.add_argument("--layout", default="aos", choices=["aos", "soa"])
    ap.add_argument("--graph", action="store_true", help="capture the K steps in one hipGraph")
    ap.add_argument("--gather-every-step", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--algorithm", default="aba", choices=["aba", "aba_tracks", "aba_lanes", "aba_chains", "aba_banks"],
                    help="lane mapping of the fused ABA: aba = the library's choice by batch size")
    ap.add_argument("--no-pipelined", action="store_true",
                    help="skip the extra (informational) measurement of two independent batches issued on two HIP streams")
    ap.add_argument("--wrenches", action="store_true", help="random external wrench on every body (as perf/runbenchmarks.jl:59-67)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist = None
        torch.cuda.set_device(0)
    device = torch.device("cuda", local_rank if world > 1 else 0)

    import rbd_amd as rbd
    from rigidbodydynamics_jl_amd import _capi

    model = rbd.load_flat_model(os.path.join(ROOT, "tests", "golden", "models", args.model + ".json"))
    B = args.batch
    tdt = torch.float64 if args.dtype == "f64" else torch.float32
    es = 8 if args.dtype == "f64" else 4
    rng = np.random.default_rng(1 + rank)
    q = rbd.rand_configuration(model, B, rng)
    v = rbd.rand_velocity(model, B, rng)
    tau = rng.random((B, model.nv))
    fext = rng.random((B, 6 * model.n_bodies)) if args.wrenches else None

    state = rbd.MechanismState(model, B, dtype=tdt, device=device, layout=args.layout)
    result = rbd.DynamicsResult(model, B, dtype=tdt, device=device, layout=args.layout)
    rbd.set_configuration_(state, q)
    rbd.set_velocity_(state, v)

    def to_dev(a):
        if a is None:
            return None
        t = torch.as_tensor(a, dtype=tdt)
        if args.layout == "soa":
            t = t.t().contiguous()
        return t.to(device)

    d_tau, d_fext = to_dev(tau), to_dev(fext)
    gathered = torch.empty((world * B, model.nv) if args.layout == "aos" else (world, model.nv, B), dtype=tdt, device=device) if world > 1 else None

    # low-overhead launch: pre-marshalled ctypes call straight into the C ABI
    L = _capi.lib()
    opts = state._opts({"aba": 0, "aba_lanes": 2, "aba_chains": 3, "aba_banks": 4, "aba_tracks": 5}[args.algorithm])
    stream = torch.cuda.current_stream(device)
    L.rbd_workspace_set_stream(state.ws.handle, ctypes.c_void_p(stream.cuda_stream))
    c_args = (state.ws.handle, B, ctypes.c_void_p(state.q.data_ptr()), ctypes.c_void_p(state.v.data_ptr()),
              ctypes.c_void_p(d_tau.data_ptr()), ctypes.c_void_p(d_fext.data_ptr() if d_fext is not None else 0),
              ctypes.c_void_p(result.vd.data_ptr()), ctypes.c_void_p(result.qd.data_ptr()), ctypes.c_void_p(0), ctypes.byref(opts))
    dyn = L.rbd_dynamics

    def step():
        st = dyn(*c_args)
        if st != 0:
            raise RuntimeError(f"rbd_dynamics status {st}: {L.rbd_status_string(st)} {L.rbd_last_hip_error()}")
        if args.gather_every_step and world > 1:
            dist.all_gather_into_tensor(gathered, result.vd)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(device)

    graph = None
    if args.graph:
        graph = torch.cuda.CUDAGraph()
        cap = torch.cuda.Stream(device)
        with torch.cuda.stream(cap):
            L.rbd_workspace_set_stream(state.ws.handle, ctypes.c_void_p(cap.cuda_stream))
            with torch.cuda.graph(graph, stream=cap):
                for _ in range(args.steps):
                    step()
        L.rbd_workspace_set_stream(state.ws.handle, ctypes.c_void_p(stream.cuda_stream))
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
            step()
    ev1.record(stream)
    torch.cuda.synchronize(device)
    if dist is not None:
        dist.barrier()
    t1 = time.perf_counter()
    wall = t1 - t0
    kernel_ms = ev0.elapsed_time(ev1) / args.steps  # average duration of one aba_kernel launch (HIP events on its stream)

    gather_ms = None
    if dist is not None:
        tw = torch.tensor([wall], dtype=torch.float64, device=device)
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        wall = float(tw.item())
        torch.cuda.synchronize(device)
        try:
            dist.all_gather_into_tensor(gathered, result.vd)  # warm-up (RCCL channel setup)
            torch.cuda.synchronize(device)
            g0 = time.perf_counter()
            dist.all_gather_into_tensor(gathered, result.vd)  # the RCCL gather of DynamicsResult.v̇ over xGMI
            torch.cuda.synchronize(device)
            gather_ms = (time.perf_counter() - g0) * 1e3
        except Exception as e:  # the gather is outside the timed region: never lose the measurement to it
            gather_ms = f"failed: {type(e).__name__}: {e}"

    # sanity: the timed work produced the right answer (checked outside the timed region, small sample)
    import oracle
    n = 32
    got = result.vd if args.layout == "aos" else result.vd.t()
    ref = oracle.dynamics(model, q[:n], v[:n], tau[:n], fext[:n] if fext is not None else None)
    err = float(np.abs(got[:n].double().cpu().numpy() - ref).max() / max(1.0, np.abs(ref).max()))
    tol = 1e-10 if args.dtype == "f64" else 3e-2
    assert err < tol, f"parity lost in bench: {err}"

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    evals = world * B * args.steps
    value = evals / wall
    alg_bytes = es * (model.nq + 3 * model.nv + (model.nq if True else 0))  # q, v, τ in; v̇ and q̇ out
    if fext is not None:
        alg_bytes += es * 6 * model.n_bodies
    achieved_gbs = alg_bytes * B / (kernel_ms * 1e-3) / 1e9
    peak_tf = FP64_VECTOR_PEAK_TF if args.dtype == "f64" else FP32_VECTOR_PEAK_TF
    achieved_tf = ABA_FLOPS_PER_EVAL * B / (kernel_ms * 1e-3) / 1e12

    traffic = None
    pmc = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if os.path.exists(pmc):
        try:
            traffic = json.load(open(pmc)).get(f"{args.model}_{args.dtype}_B{B}")
        except Exception:
            traffic = None

    out = {
        "metric": "ABA dynamics! evals/sec (Atlas 30-DoF, batch)", "value": value, "unit": "evals/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"{args.model} nq={model.nq} nv={model.nv} bodies={model.n_bodies}, batch={B}/GPU, "
                               f"{args.dtype} fused ABA dynamics! (BASELINE configs[1])",
                   "batch_per_gpu": B, "layout": args.layout, "external_wrenches": bool(args.wrenches),
                   "hip_graph": bool(args.graph), "parallelism": f"batch-sharded x{world}, no data-path collective",
                   "gather_every_step": bool(args.gather_every_step)},
        "roofline": {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic,
                     "kernel": (L.rbd_workspace_last_kernel(state.ws.handle) or b"").decode(), "kernel_ms": kernel_ms, "algorithmic_bytes_per_eval": alg_bytes},
        "alu": {"bound": "fp64 vector ALU" if args.dtype == "f64" else "fp32 vector ALU", "achieved": achieved_tf, "peak": peak_tf,
                "unit": "TFLOP/s", "frac": achieved_tf / peak_tf, "flops_per_eval": ABA_FLOPS_PER_EVAL,
                "note": "the path is ALU/latency bound, not HBM bound (SURVEY.md F8): compulsory traffic is ~1.5 KB/eval"},
        "parity_rel_err_vs_oracle": err,
        "published_reference": {"dynamics!_us_per_eval": 9.874, "evals_per_s": 1.01e5, "hardware": "Apple M2, 1 thread, Julia 1.11",
                                "source": "docs/src/benchmarks.md:71-78"},
    }
    if gather_ms is not None:
        out["rccl_all_gather_vdot_ms"] = gather_ms

    if not args.no_pipelined and world == 1:
        # Informational, NOT `value`: the same step count with two INDEPENDENT batches of B states alternating on two HIP streams
        # (one workspace each).  At B = 4096 a single launch leaves every SIMD with one wavefront (latency-bound, VALU busy ~36 %);
        # a second launch in flight gives each SIMD a second wavefront to interleave.  Callers whose batches do not depend on each
        # other (sampling-based control, the reference's own benchmark loop) can run this way; `simulate` cannot.
        try:
            rng2 = np.random.default_rng(1001)
            state2 = rbd.MechanismState(model, B, dtype=tdt, device=device, layout=args.layout)
            result2 = rbd.DynamicsResult(model, B, dtype=tdt, device=device, layout=args.layout)
            rbd.set_configuration_(state2, rbd.rand_configuration(model, B, rng2))
            rbd.set_velocity_(state2, rbd.rand_velocity(model, B, rng2))
            s_a, s_b = torch.cuda.Stream(device), torch.cuda.Stream(device)
            L.rbd_workspace_set_stream(state.ws.handle, ctypes.c_void_p(s_a.cuda_stream))
            L.rbd_workspace_set_stream(state2.ws.handle, ctypes.c_void_p(s_b.cuda_stream))
            opts2 = state2._opts({"aba": 0, "aba_lanes": 2, "aba_chains": 3, "aba_banks": 4, "aba_tracks": 5}[args.algorithm])
            c_args2 = (state2.ws.handle, B, ctypes.c_void_p(state2.q.data_ptr()), ctypes.c_void_p(state2.v.data_ptr()),
                       ctypes.c_void_p(d_tau.data_ptr()), ctypes.c_void_p(d_fext.data_ptr() if d_fext is not None else 0),
                       ctypes.c_void_p(result2.vd.data_ptr()), ctypes.c_void_p(result2.qd.data_ptr()), ctypes.c_void_p(0), ctypes.byref(opts2))
            both = (c_args, c_args2)
            for k in range(args.warmup):
                dyn(*both[k & 1])
            torch.cuda.synchronize(device)
            p0 = time.perf_counter()
            for k in range(args.steps):
                dyn(*both[k & 1])
            torch.cuda.synchronize(device)
            p1 = time.perf_counter()
            L.rbd_workspace_set_stream(state.ws.handle, ctypes.c_void_p(stream.cuda_stream))
            out["pipelined_independent_batches"] = {"streams": 2, "value": B * args.steps / (p1 - p0), "unit": "evals/s",
                                                    "ms_per_step": (p1 - p0) / args.steps * 1e3,
                                                    "note": "informational: two independent batches in flight; not the headline value"}
        except Exception as e:  # never lose the headline line to the extra measurement
            out["pipelined_independent_batches"] = f"failed: {type(e).__name__}: {e}"

    if not args.no_cpu_baseline and world == 1:  # the CPU leg is reported at N = 1 only (rank 0 of a multi-GPU run would time it while its peers wait)
        # host threads actually available to this process: affinity mask, capped by the cgroup CPU quota when there is one
        ncores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        try:
            quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
            if quota != "max":
                ncores = max(1, min(ncores, int(round(int(quota) / int(period)))))
        except Exception:
            pass
        os.environ.setdefault("OMP_PROC_BIND", "close")
        S1 = 2048
        qs, vs, ts = q[:S1], v[:S1], tau[:S1]
        fs = fext[:S1] if fext is not None else None
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
        fN = np.tile(fext, (tile, 1)) if fext is not None else None
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
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
