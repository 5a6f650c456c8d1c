"""Where the compiled walk kernel overtakes the banked one: dynamics! / inverse_dynamics! between one and two resident rounds of the banked kernel (4096 .. 16384 Atlas
states), both precisions, graph-replayed µs per launch.  RBD_SPEC_WALK_MIN_BATCH=1 so that the forced walk mapping is the compiled kernel at every size."""
import os, sys
os.environ.setdefault("RBD_SPEC_WALK_MIN_BATCH", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import rbd_amd as rbd
model = rbd.load_flat_model(os.path.join(ROOT, "tests", "golden", "models", "atlas_floating.json"))
rng = np.random.default_rng(5)
def timed(f):
    for _ in range(5): f()
    torch.cuda.synchronize()
    cap = torch.cuda.Stream(); g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(cap):
        f()
        with torch.cuda.graph(g, stream=cap):
            for _ in range(10): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g.replay(); torch.cuda.synchronize()
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) * 1000 / 50, 2)
for dt in (torch.float64, torch.float32):
    for B in (2048, 3072, 4096, 4352, 5120, 6144, 8192, 12288, 16384):
        state = rbd.MechanismState(model, B, dtype=dt); result = rbd.DynamicsResult(model, B, dtype=dt)
        rbd.set_configuration_(state, rbd.rand_configuration(model, B, rng)); rbd.set_velocity_(state, rbd.rand_velocity(model, B, rng))
        tau = torch.rand(B, model.nv, dtype=dt, device="cuda"); out = torch.zeros_like(tau)
        row = {}
        for alg, mp in (("aba_banks", "banks"), ("aba_walk", "walk"), ("aba", "auto")):
            row[alg] = timed(lambda: rbd.dynamics_(result, state, tau, algorithm=alg)); ka = rbd.last_kernel(state).split(" ")[0]
            row["id_" + mp] = timed(lambda: rbd.inverse_dynamics_(out, state, tau, mapping=mp)); kr = rbd.last_kernel(state).split(" ")[0]
            row["k_" + alg] = ka + "/" + kr
        print(str(dt).split(".")[1], "B", B, "dynamics! banks %.1f walk %.1f auto %.1f | inverse_dynamics! banks %.1f walk %.1f auto %.1f | auto: %s" %
              (row["aba_banks"], row["aba_walk"], row["aba"], row["id_banks"], row["id_walk"], row["id_auto"], row["k_aba"]), flush=True)
