"""Time the bf16 mode's code search stand-alone (A/B of kernel variants on one box): python scripts/one_vq.py [N D K]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_sound_generation_amd import ops
N, D, K = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (655360, 128, 512)
dev = "cuda:0"
x = torch.randn(N, D, device=dev)
e = torch.randn(K, D, device=dev) * 0.5
for _ in range(3):
    out = ops.vq_forward(x, e, want_codes=False, impl="bf16x3", codes_bf16="relu")
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    out = ops.vq_forward(x, e, want_codes=False, impl="bf16x3", codes_bf16="relu")
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 20
print(f"vq_forward_bf16x3 N={N} D={D} K={K}: {us:.1f} us  {2.0 * N * D * K / us * 1e-6:.1f} TFLOP/s algorithmic")
