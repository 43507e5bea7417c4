"""Measured streaming bandwidth of the box's MI355X next to the 8 TB/s spec figure (SURVEY 8d asks
for both): device copy and triad over fp64 arrays far larger than the 256 MB Infinity Cache.
Plumbing only (torch elementwise kernels); bytes counted as read + written."""
import time
import torch
assert torch.cuda.is_available()
n = 1 << 28                                    # 2 GiB per fp64 array
a = torch.empty(n, dtype=torch.float64, device="cuda")
b = torch.rand(n, dtype=torch.float64, device="cuda")
c = torch.rand(n, dtype=torch.float64, device="cuda")
def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps
t_copy = timed(lambda: a.copy_(b))
t_triad = timed(lambda: torch.add(b, c, alpha=1.5, out=a))
t_read = timed(lambda: b.sum())
p = torch.cuda.get_device_properties(0)
print("HBM", p.name, "CUs", p.multi_processor_count, "copy %.2f TB/s" % (2 * 8 * n / t_copy / 1e12),
      "triad %.2f TB/s" % (3 * 8 * n / t_triad / 1e12), "read-only %.2f TB/s" % (8 * n / t_read / 1e12))
