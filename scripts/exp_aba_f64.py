"""Round-6 experiment: dynamics! in fp64 for mechanisms with 3-dof / inner 6-dof joints through the lane-per-state program in doubles (aba_spec_f64) against the
one-body-per-lane kernel, with parity against the oracle.  usage: python scripts/exp_aba_f64.py [randmech1] [65536]"""
import os, sys, json
os.environ.setdefault("RBD_JIT_ASYNC", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import rbd_amd as rbd, oracle
name = sys.argv[1] if len(sys.argv) > 1 else "randmech1"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
model = rbd.flatten(rbd.randmech(np.random.default_rng(int(name[8:] or 1))))
rng = np.random.default_rng(3)
q, v, tau = rbd.rand_configuration(model, B, rng), rbd.rand_velocity(model, B, rng), rng.random((B, model.nv))
fe = rng.random((B, 6 * model.n_bodies))
state = rbd.MechanismState(model, B); result = rbd.DynamicsResult(model, B)
rbd.set_configuration_(state, q); rbd.set_velocity_(state, v)
d_tau, d_fe = torch.as_tensor(tau).cuda(), torch.as_tensor(fe).cuda()
ref = oracle.dynamics(model, q[:4096], v[:4096], tau[:4096], nthreads=8)
ref_fe = oracle.dynamics(model, q[:4096], v[:4096], tau[:4096], fe[:4096], nthreads=8)
out = {}
for algo in ("aba", "aba_compiled"):
    for wr in (False, True):
        f = lambda: rbd.dynamics_(result, state, d_tau, d_fe if wr else None, algorithm=algo)
        try:
            for _ in range(3): f()
        except Exception as e:
            out[f"{algo}{'+fext' if wr else ''}"] = str(e); continue
        torch.cuda.synchronize()
        got = result.vd[:4096].cpu().numpy()
        r = ref_fe if wr else ref
        err = float(np.abs(got - r).max() / max(1.0, np.abs(r).max()))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        out[f"{algo}{'+fext' if wr else ''}"] = {"us": round(e0.elapsed_time(e1) / 20 * 1e3, 1), "err": err, "kernel": rbd.last_kernel(state)[:24]}
print(json.dumps({"model": name, "B": B, "nv": model.nv, "results": out}))
