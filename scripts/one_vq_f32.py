"""The exact (fp32) search alone at BASELINE configs[3]'s and configs[1]'s shapes."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_sound_generation_amd import ops
dev = "cuda:0"
for N, D, K in ((81920, 256, 8192), (655360, 128, 512)):
    x = torch.randn(N, D, device=dev) * 0.05
    e = (torch.rand(K, D, device=dev) * 2 - 1) / K
    for _ in range(2): ops.vq_forward(x, e, want_codes=True)
    torch.cuda.synchronize()
    ts = []
    for _ in range(8):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); ops.vq_forward(x, e, want_codes=True); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    us = float(np.median(ts))
    print(f"N={N} D={D} K={K}: {us:.1f} us = {2.0 * N * K * D / us / 1e6:.1f} TFLOP/s = {2.0 * N * K * D / us / 1e6 / 157.3:.3f} of the fp32 MFMA peak")
