"""Run one bf16 conv layer of the training step a few times (for rocprofv3 --pmc / --kernel-trace):
python scripts/one_conv.py {3x3|4x4s2|convT} [iters]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_sound_generation_amd import ops
kind = sys.argv[1] if len(sys.argv) > 1 else "3x3"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev, DT = "cuda:0", torch.bfloat16
B, D = int(os.environ.get("B", "128")), int(os.environ.get("D", "128"))
k, s_, p_, ih, iw, tr = {"3x3": (3, 1, 1, 20, 256, False), "4x4s2": (4, 2, 1, 40, 512, False), "convT": (4, 2, 1, 20, 256, True)}[kind]
d = ops.conv_desc(B, ih, iw, D, D, k, s_, p_, transposed=tr, dtype=DT)
x = torch.relu(torch.randn(B, ih, iw, D, device=dev)).to(DT)
w = torch.randn(D, D, k, k, device=dev) * 0.05
wf, wd = ops.pack_weights(d, w)
bias = torch.zeros(D, device=dev)
for _ in range(iters):
    y = ops.conv_forward(d, x, wf, bias)
torch.cuda.synchronize()
if os.environ.get("TIME", "0") == "1":       # stand-alone timing (cold caches differ from the step's: an A/B tool, not a benchmark)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        y = ops.conv_forward(d, x, wf, bias)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print(f"{kind}: {us:.1f} us  {2.0 * y.numel() * D * k * k / (4 if tr else 1) / us * 1e-6:.1f} TFLOP/s")
print("ok", float(y.float().abs().mean()))
