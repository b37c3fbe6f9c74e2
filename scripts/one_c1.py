"""Run the fused input layer (forward + backward, bf16) a few times at BASELINE configs[1]'s extent, for rocprofv3 --kernel-trace --stats:
python scripts/one_c1.py [iters]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_sound_generation_amd import ops
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = "cuda:0"
B, C = int(os.environ.get("B", "128")), int(os.environ.get("D", "128"))
H, W = 80, 1024
img = torch.randn(B, H, W, device=dev) * 0.7 + 0.3
w = torch.randn(C, 1, 4, 4, device=dev) * 0.3
b = torch.randn(C, device=dev) * 0.2
gamma, beta = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.2
rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
dy = torch.randn(B, H // 2, W // 2, C, device=dev).bfloat16()
for _ in range(iters):
    y, mean, invstd = ops.c1conv_bn_relu_forward(img, w, b, gamma, beta, rm, rv, training=True, out_dtype=torch.bfloat16)
    dw, dbias, dg, db = ops.c1conv_bn_relu_backward(img, w, b, gamma, beta, mean, invstd, dy)
torch.cuda.synchronize()
print("ok", float(dw.abs().mean()))
