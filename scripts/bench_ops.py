"""Throughput of every hot-path entry point (not the headline line — bench.py is): evals/s and µs per launch, graph-replayed so
the GPU is the bottleneck.  usage: python scripts/bench_ops.py [--batch 4096] [--dtype f64] [--model atlas_floating]"""
import argparse, json, os, sys
os.environ.setdefault("RBD_JIT_ASYNC", "0")  # wait for the kernels compiled per mechanism instead of starting on the interpreting ones
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import rbd_amd as rbd

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4096); ap.add_argument("--dtype", default="f64"); ap.add_argument("--model", default="atlas_floating")
ap.add_argument("--no-sim", action="store_true"); ap.add_argument("--reps", type=int, default=200); ap.add_argument("--only", default="", help="substring filter on the op names")
args = ap.parse_args()
tdt = torch.float64 if args.dtype == "f64" else torch.float32
if args.model.startswith("randmech"):  # the reference's own test mechanism (test/test_mechanism_algorithms.jl:1-11; SPQuatFloating -> QuaternionSpherical), seed = the suffix
    model = rbd.flatten(rbd.randmech(np.random.default_rng(int(args.model[8:] or 1))))
else:
    model = rbd.load_flat_model(os.path.join(ROOT, "tests", "golden", "models", args.model + ".json"))
B = args.batch
rng = np.random.default_rng(1)
if args.no_sim: pass
state = rbd.MechanismState(model, B, dtype=tdt); result = rbd.DynamicsResult(model, B, dtype=tdt)
rbd.set_configuration_(state, rbd.rand_configuration(model, B, rng)); rbd.set_velocity_(state, rbd.rand_velocity(model, B, rng))
tau = torch.rand(B, model.nv, dtype=tdt, device="cuda"); out = torch.zeros_like(tau)
Abuf = torch.zeros(B, 6 * model.nv, dtype=tdt, device="cuda")
jwbuf = torch.zeros(B, 6 * model.n_bodies, dtype=tdt, device="cuda"); accbuf = torch.zeros_like(jwbuf)
result_b = rbd.DynamicsResult(model, B, dtype=tdt, bodies=True)
ops = {
    "dynamics! (ABA)": lambda: rbd.dynamics_(result, state, tau),
    "dynamics! (CRBA+Cholesky route)": lambda: rbd.dynamics_(result, state, tau, algorithm="crba"),
    "inverse_dynamics!": lambda: rbd.inverse_dynamics_(out, state, tau),
    "inverse_dynamics! with jointwrenches + accelerations (the reference benchmark's call)": lambda: rbd.inverse_dynamics_(out, state, tau, jointwrenchesout=jwbuf, accelerations=accbuf),
    "dynamics! into a DynamicsResult with per-body fields": lambda: rbd.dynamics_(result_b, state, tau),
    "dynamics_bias!": lambda: rbd.dynamics_bias_(result, state),
    "mass_matrix!": lambda: rbd.mass_matrix_(result, state),
    "mass_matrix! + Cholesky solve": lambda: rbd.mass_matrix_solve_(out, state, tau, result.massmatrix),
    "M^-1 rhs via the articulated-body solve": lambda: rbd.mass_matrix_solve_(out, state, tau, algorithm="aba"),
    "momentum_matrix!": lambda: rbd.momentum_matrix_(Abuf, state),
    "geometric_jacobian! (root -> last body)": lambda: rbd.geometric_jacobian_(Abuf, state, -1, model.n_bodies - 1),
    "center_of_mass": lambda: rbd.center_of_mass(state),
    "kinetic_energy": lambda: rbd.kinetic_energy(state),
    "momentum + momentum_rate_bias": lambda: rbd.momentum(state),
}
res = {}
for name, f in ops.items():
    if args.only and args.only not in name: continue
    for _ in range(3): f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); cap = torch.cuda.Stream()
    with torch.cuda.stream(cap):
        f()
        with torch.cuda.graph(g, stream=cap):
            for _ in range(args.reps): f()
    torch.cuda.synchronize()
    for _ in range(3): g.replay()  # untimed: clocks up, caches warm (the first op of a fresh process otherwise reads ~20 % high)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / args.reps * 1e3
    res[name] = {"us_per_launch": round(us, 2), "Mevals_per_s": round(B / us, 1), "kernel": rbd.last_kernel(state)[:60]}
if args.no_sim:
    print(json.dumps({"model": args.model, "batch": B, "dtype": args.dtype, "ops": res})); sys.exit(0)
# simulate: steps/s (each step = 4 dynamics! + 5 stage kernels)
q0, v0 = state.q.clone(), state.v.clone()
rbd.simulate_(state, 0.0095, dt=1e-3); torch.cuda.synchronize()
state.q.copy_(q0); state.v.copy_(v0)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); rbd.simulate_(state, 0.0995, dt=1e-3); e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 100 * 1e3
res["simulate (RK4 step = 4 dynamics!)"] = {"us_per_step": round(us, 2), "Msteps_per_s": round(B / us, 2)}
print(json.dumps({"model": args.model, "batch": B, "dtype": args.dtype, "ops": res}))
