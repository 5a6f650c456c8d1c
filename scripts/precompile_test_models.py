"""Compiles the run-time specialised kernels of every mechanism the GPU tests use into the library's cache (csrc/jit_cache/, no GPU needed), so that the
GPU box only loads them.  Optional: anything missing is compiled there on first use.  hiprtc compiles one program at a time per process: the mechanisms are
shared out over one process per core (`python scripts/precompile_test_models.py` starts them; `... k n` is worker k of n)."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
if len(sys.argv) < 3:
    n = max(1, len(os.sched_getaffinity(0)))
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), str(k), str(n)]) for k in range(n)]
    # ... and the programs that are WRONG by construction (RBD_TUNE spec_variant=16: the passes on made-up rows) which
    # tests/test_state_kernels.py::test_first_use_check_drops_a_wrong_program expects the library to catch
    procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "wrong", "1"], env=dict(os.environ, RBD_TUNE="spec_variant=16")))
    sys.exit(max(p.wait() for p in procs))
import torch
import rbd_amd as rbd
from conftest import build_models
models = build_models(rbd)
if sys.argv[1] == "wrong":
    jobs, k, n = [("double_pendulum", torch.float32), ("inner_floating", torch.float64)], 0, 1
else:
    k, n = int(sys.argv[1]), int(sys.argv[2])
    jobs = [(name, dt) for name in sorted(models) for dt in (torch.float32, torch.float64)]
for name, dt in jobs[k::n]:
    t = time.time()
    ok, log = rbd.jit_precompile(models[name], dt)
    refused = [l for l in log.splitlines() if "not used" in l]  # (a walk program the allocator gave accumulation registers or scratch: the interpreting kernel serves instead)
    print(name, dt, ok, round(time.time() - t, 1), "s", ("  REFUSED: " + " | ".join(refused)) if refused else "", flush=True)
