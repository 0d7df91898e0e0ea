"""Diagnostic: what this box's HBM sustains for a pure fill (write) and a copy (read + write) through torch -- the practical ceiling the
observation kernel's 16 G^2 bytes per render are measured against."""
import torch
dev = torch.device("cuda:0")
n = 1 << 28                                                 # 1 GiB of float32
x = torch.empty(n, dtype=torch.float32, device=dev); y = torch.empty_like(x)
def timed(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
t = timed(lambda: x.zero_()); print("fill  1 GiB: %.3f ms  %.2f TB/s written" % (t * 1e3, 4 * n / t / 1e12))
t = timed(lambda: y.copy_(x)); print("copy  1 GiB: %.3f ms  %.2f TB/s read + written" % (t * 1e3, 8 * n / t / 1e12))
