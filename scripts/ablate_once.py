"""Times aba_kernel with the RBD_ABA_STOP_AFTER early exit (profiling aid): which phase costs what."""
import ctypes, json, os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "oracle"))
import numpy as np, torch
import rbd_amd as rbd
from rigidbodydynamics_jl_amd import _capi
ph = os.environ.get("RBD_ABA_STOP_AFTER", "0")
res = {}
L = _capi.lib()
for dt, tdt in (("f64", torch.float64), ("f32", torch.float32)):
    for B in (512, 4096, 65536):
        model = rbd.load_flat_model("tests/golden/models/atlas_floating.json")
        rng = np.random.default_rng(1)
        state = rbd.MechanismState(model, B, dtype=tdt); result = rbd.DynamicsResult(model, B, dtype=tdt)
        rbd.set_configuration_(state, rbd.rand_configuration(model, B, rng)); rbd.set_velocity_(state, rbd.rand_velocity(model, B, rng))
        tau = torch.rand(B, model.nv, dtype=tdt, device="cuda")
        opts = state._opts()
        args = (state.ws.handle, B, ctypes.c_void_p(state.q.data_ptr()), ctypes.c_void_p(state.v.data_ptr()), ctypes.c_void_p(tau.data_ptr()),
                ctypes.c_void_p(0), ctypes.c_void_p(result.vd.data_ptr()), ctypes.c_void_p(result.qd.data_ptr()), ctypes.c_void_p(0), ctypes.byref(opts))
        n = 400 if B <= 4096 else 50
        g = torch.cuda.CUDAGraph()
        cap = torch.cuda.Stream()
        with torch.cuda.stream(cap):
            L.rbd_workspace_set_stream(state.ws.handle, ctypes.c_void_p(cap.cuda_stream))
            for _ in range(5): assert L.rbd_dynamics(*args) == 0
            with torch.cuda.graph(g, stream=cap):
                for _ in range(n): assert L.rbd_dynamics(*args) == 0
        torch.cuda.synchronize()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        res[f"{dt}_B{B}"] = round(e0.elapsed_time(e1) / n * 1e3, 2)
print(json.dumps({"stop": int(ph), **res}))
