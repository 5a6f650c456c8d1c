"""What the HBM takes on this box: 340 MB (configs[2]'s M for 65 536 states) filled, zeroed and copied by the stock element-wise kernels — the ceiling emit_spec is read against."""
import torch
x = torch.empty(65536 * 36 * 36, dtype=torch.float32, device="cuda")
y = torch.empty_like(x)
def t(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n
mb = x.numel() * 4 / 1e6
for name, f in (("fill_", lambda: x.fill_(1.0)), ("zero_", lambda: x.zero_()), ("copy_ (read+write)", lambda: y.copy_(x))):
    us = t(f)
    print(name, round(us, 1), "us for", round(mb), "MB:", round(mb / us / 1e6 * 1e6 / 1e6, 2), "TB/s (copy_: the same again read)")
