"""N > 1 path on CPU: two gloo processes shard a batch, each evaluates its shard, the shards are gathered and compared
with the single-process result.  The compute stand-in on CPU is the oracle (test infrastructure) — the product kernels
need a GPU; what is under test here is the sharding and the gather (rigidbodydynamics.jl_amd/distributed.py)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, B, dst, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    import rbd_amd as rbd
    model = rbd.load_flat_model(os.path.join(ROOT, "tests", "golden", "models", "atlas_floating.json"))
    rng = np.random.default_rng(5)  # same full batch on every rank; each evaluates only its shard
    q = rbd.rand_configuration(model, B, rng)
    v = rbd.rand_velocity(model, B, rng)
    tau = rng.random((B, model.nv))
    lo, hi = rbd.shard_range(B, rank, world)
    local = torch.from_numpy(oracle.dynamics(model, q[lo:hi], v[lo:hi], tau[lo:hi]))
    full = rbd.gather_results(local, B, dst=dst)
    if dst is None or rank == dst:
        ref = oracle.dynamics(model, q, v, tau)
        ok = full is not None and tuple(full.shape) == (B, model.nv) and np.array_equal(full.numpy(), ref)
    else:
        ok = full is None
    open(os.path.join(out_dir, f"ok_{rank}"), "w").write("1" if ok else "0")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("B,dst", [(64, None), (37, None), (37, 0), (5, 1)])
def test_two_rank_shard_and_gather(tmp_path, B, dst):
    world = 2
    port = 29500 + (os.getpid() + B) % 2000
    mp.spawn(_worker, args=(world, port, B, dst, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert open(tmp_path / f"ok_{r}").read() == "1"


def test_shard_ranges_cover_batch():
    sys.path.insert(0, ROOT)
    import rbd_amd as rbd
    for B in (0, 1, 7, 8, 4096, 524288):
        for world in (1, 2, 3, 8):
            rs = [rbd.shard_range(B, r, world) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == B
            assert all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in rs]
            assert max(sizes) - min(sizes) <= 1


def test_bench_launcher_spawns_the_ranks_it_is_asked_for():
    """`python bench.py --gpus N` must start N ranks itself (it used to run one process whatever N said): the launcher path, on gloo."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--selftest-launch"], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 2 and rec["ranks_counted"] == 2 and rec["asked"] == 2


def test_bench_refuses_more_gpus_than_visible():
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "64"], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 2
    assert "refusing" in json.loads(out.stdout.strip().splitlines()[-1])["error"]
