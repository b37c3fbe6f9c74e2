"""Isolated timings of the four patch_gemm launch kinds of the training step (post-ReLU-like data), median of 30."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_sound_generation_amd import ops
dev = "cuda:0"
B, D = int(os.environ.get("B", "128")), int(os.environ.get("D", "128"))
DT = torch.bfloat16
big = torch.randn(B, 40, 512, D, device=dev).to(DT)
out = []
for name, (k, s_, ih, iw, tr, role) in {"3x3 fwd": (3, 1, 20, 256, False, "f"), "3x3 dgrad+add+mask": (3, 1, 20, 256, False, "d"),
                                        "4x4/s2 fwd": (4, 2, 40, 512, False, "f"), "convT fwd": (4, 2, 20, 256, True, "f"), "convT dgrad+mask": (4, 2, 40, 512, False, "m")}.items():
    d = ops.conv_desc(B, ih, iw, D, D, k, s_, 1, transposed=tr, dtype=DT)
    x = torch.relu(torch.randn(B, ih, iw, D, device=dev)).to(DT)
    w = torch.randn(D, D, k, k, device=dev) * 0.05
    wf, wd = ops.pack_weights(d, w); bias = torch.zeros(D, device=dev)
    dy = (torch.randn(B, d.OH, d.OW, D, device=dev) * (torch.rand(B, d.OH, d.OW, D, device=dev) > 0.5)).to(DT)
    skip = torch.randn(B, ih, iw, D, device=dev).to(DT)
    def run():
        if role == "f": return ops.conv_forward(d, x, wf, bias)
        if role == "d": return ops.conv_dgrad(d, dy, wd, add=skip, relu_x=x)
        return ops.conv_dgrad(d, dy, wd, relu_x=x)
    ts = []
    for i in range(35):
        ops.convert(big, DT, out=big)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    out.append(f"{name} {np.median(ts[5:]):.1f}")
print(" | ".join(out))
