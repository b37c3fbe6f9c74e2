"""One conv shape launched N times (for rocprofv3 --pmc passes).  usage: conv_only.py <3x3|3x3d|s2|convT> [n]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_sound_generation_amd import ops
dev = "cuda:0"
which = sys.argv[1] if len(sys.argv) > 1 else "3x3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
B, D = int(os.environ.get("B", "128")), int(os.environ.get("D", "128"))
DT = torch.bfloat16
k, s_, ih, iw, tr = {"3x3": (3, 1, 20, 256, False), "3x3d": (3, 1, 20, 256, False), "s2": (4, 2, 40, 512, False), "convT": (4, 2, 20, 256, True)}[which]
d = ops.conv_desc(B, ih, iw, D, D, k, s_, 1, transposed=tr, dtype=DT)
x = torch.relu(torch.randn(B, ih, iw, D, device=dev)).to(DT)
w = torch.randn(D, D, k, k, device=dev) * 0.05
wf, wd = ops.pack_weights(d, w)
bias = torch.zeros(D, device=dev)
dy = torch.randn(B, d.OH, d.OW, D, device=dev).to(DT)
skip = torch.randn(B, ih, iw, D, device=dev).to(DT)
for _ in range(n):
    if which == "3x3d":
        ops.conv_dgrad(d, dy, wd, add=skip, relu_x=x)
    else:
        ops.conv_forward(d, x, wf, bias)
torch.cuda.synchronize()
