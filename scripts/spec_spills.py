"""Registers and spills of the kernels compiled per mechanism, for every mechanism of the tests (hiprtc, no GPU needed): a one-lane-per-state kernel that
spills ANYTHING is not picked by the library on its own (rbd_capi.hip), so this is the list of who gets the fast path.
usage: python scripts/spec_spills.py [name ...]"""
import os, re, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
if len(sys.argv) >= 3 and sys.argv[1] == "--worker":
    import torch
    import rbd_amd as rbd
    from conftest import build_models
    name = sys.argv[2]
    m = build_models(rbd)[name]
    for dt, fams in ((torch.float32, (0, 1, 2)), (torch.float64, (0, 2))):
        for fam in fams:
            while True:
                st = rbd.jit_status(m, dt, fam)
                if st != 0: break
                time.sleep(0.2)
    out = []
    for f in sorted(os.listdir(os.environ["RBD_JIT_CACHE"])):
        notes = subprocess.run([READELF, "--notes", os.path.join(os.environ["RBD_JIT_CACHE"], f)], capture_output=True, text=True).stdout
        for k in re.finditer(r"\.agpr_count:\s+(\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)", notes, re.S):
            if k.group(2).startswith(("aba_spec", "rnea_spec", "crba_spec")):
                out.append(f"{k.group(2)}: regs {k.group(4)} scratch {k.group(3)} B spills {k.group(5)}")
    print(f"{name:22s} " + " | ".join(sorted(out)), flush=True)
    sys.exit(0)
import rbd_amd as rbd
from conftest import build_models
names = sys.argv[1:] or sorted(build_models(rbd))
procs = []
for name in names:
    d = tempfile.mkdtemp(prefix="rbd_spills_"); os.chmod(d, 0o700)
    procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", name], env=dict(os.environ, RBD_JIT_CACHE=d)))
    while sum(p.poll() is None for p in procs) >= max(1, len(os.sched_getaffinity(0))): time.sleep(0.2)
for p in procs: p.wait()
