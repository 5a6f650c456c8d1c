"""Timing of the one-lane-per-state kernels against the lane-per-body ones over the batch size (Atlas floating):
mass_matrix!, dynamics_bias!, mass_matrix! + Cholesky solve, dynamics! by the reference's route and by the fused ABA.
usage: python scripts/state_sweep.py [f32|f64] [aos|soa]   (RBD_TUNE=state_min_batch=<n> is set per measurement)"""
import json, os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def child(dt, layout, B):
    import numpy as np, torch
    import rbd_amd as rbd
    tdt = torch.float32 if dt == "f32" else torch.float64
    model = rbd.load_flat_model(os.path.join(ROOT, "tests/golden/models/atlas_floating.json"))
    rng = np.random.default_rng(1)
    state = rbd.MechanismState(model, B, dtype=tdt, layout=layout)
    result = rbd.DynamicsResult(model, B, dtype=tdt, layout=layout)
    rbd.set_configuration_(state, rbd.rand_configuration(model, B, rng)); rbd.set_velocity_(state, rbd.rand_velocity(model, B, rng))
    tau = torch.rand_like(state.v); x = torch.zeros_like(state.v)
    def t(fn, n):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return round(e0.elapsed_time(e1) / n * 1e3, 1)
    n = 200 if B <= 8192 else 40
    out = {"B": B}
    out["mass_matrix_us"] = t(lambda: rbd.mass_matrix_(result, state), n)
    out["bias_us"] = t(lambda: rbd.dynamics_bias_(result, state), n)
    out["mm_solve_us"] = t(lambda: rbd.mass_matrix_solve_(x, state, tau, result.massmatrix), n)
    out["dyn_crba_us"] = t(lambda: rbd.dynamics_(result, state, tau, algorithm="crba"), n)
    out["dyn_aba_us"] = t(lambda: rbd.dynamics_(result, state, tau), n)
    print(json.dumps(out))

if __name__ == "__main__":
    if len(sys.argv) > 3 and sys.argv[1] == "--child":
        child(sys.argv[2], sys.argv[3], int(sys.argv[4]))
        sys.exit(0)
    dt = sys.argv[1] if len(sys.argv) > 1 else "f32"
    layout = sys.argv[2] if len(sys.argv) > 2 else "aos"
    for B in (4096, 16384, 32768, 65536, 131072, 262144):
        for mode, minb in (("lane-per-body", str(1 << 40)), ("lane-per-state", "1")):
            env = dict(os.environ, RBD_TUNE="state_min_batch=" + minb)
            r = subprocess.run([sys.executable, __file__, "--child", dt, layout, str(B)], env=env, capture_output=True, text=True)
            line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]
            print(dt, layout, mode, line, flush=True)
