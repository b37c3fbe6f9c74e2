"""What simple streaming kernels of this library reach on this box (the practical ceiling for the HBM-bound half of the step):
convert (1 read + 1 write), bn_apply (2 reads + 1 write), bn_backward_sums (2 reads), at the step's tensor sizes."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_sound_generation_amd import ops
dev = "cuda:0"
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    return float(np.median(ts))
C = 128
for name, M in (("low-res (168 MB bf16)", 128 * 20 * 256), ("high-res (671 MB bf16)", 128 * 40 * 512)):
    x = torch.randn(M, C, device=dev).to(torch.bfloat16)
    y = torch.empty_like(x)
    r = torch.randn(M, C, device=dev).to(torch.bfloat16)
    nb = x.numel() * 2
    us = timeit(lambda: ops.convert(x, torch.bfloat16, out=y, relu=True))
    print(f"{name}: convert+relu {us:7.1f} us = {2 * nb / us / 1e6:5.2f} TB/s", end="; ")
    mean, invstd = ops.bn_stats(x, C)
    g = torch.ones(C, device=dev); b = torch.zeros(C, device=dev)
    us = timeit(lambda: ops.bn_apply(x, mean, invstd, g, b, relu=False, residual=r, out=y))
    print(f"bn_apply+residual {us:7.1f} us = {3 * nb / us / 1e6:5.2f} TB/s", end="; ")
    us = timeit(lambda: ops.bn_stats(x, C))
    print(f"bn_stats {us:7.1f} us = {nb / us / 1e6:5.2f} TB/s", end="; ")
    us = timeit(lambda: ops.bn_backward_sums(x, r, mean, invstd, g))
    print(f"bn_backward_sums {us:7.1f} us = {2 * nb / us / 1e6:5.2f} TB/s")
