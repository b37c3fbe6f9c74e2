"""Both nearest-code searches at several (D, K): the time that does not scale with K is the per-block prologue + epilogue."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_sound_generation_amd import ops
dev = "cuda:0"
for impl, peak in (("mfma", 157.3), ("bf16x3", 2500.0)):
    for N, D, K in ((655360, 128, 512), (655360, 128, 1024), (655360, 128, 2048), (81920, 256, 8192), (81920, 256, 4096)):
        x = torch.randn(N, D, device=dev) * 0.05
        e = (torch.rand(K, D, device=dev) * 2 - 1) / K
        kw = dict(want_codes=False, impl=impl)
        if impl == "bf16x3": kw["codes_bf16"] = "relu"
        for _ in range(2): ops.vq_forward(x, e, **kw)
        torch.cuda.synchronize()
        ts = []
        for _ in range(6):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); ops.vq_forward(x, e, **kw); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3)
        us = float(np.median(ts))
        print(f"{impl:7s} N={N} D={D} K={K}: {us:8.1f} us = {2.0 * N * K * D / us / 1e6:7.1f} TF algorithmic = {2.0 * N * K * D / us / 1e6 / peak:.3f} of peak")
