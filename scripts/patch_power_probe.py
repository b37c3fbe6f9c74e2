"""Is patch_gemm bound by the clock the chip holds under load (power), not by its instruction stream?  Times the SAME launch
(3x3 conv forward, B = 128, D = 128, bf16) on operands of different toggle density: all-zero, post-ReLU-like (half zeros),
dense random; back to back (sustained) and alternating with an HBM-bound kernel as inside the training step.  Equal cycle counts
whatever the data (the MFMA takes the same cycles on zeros), so any difference in TFLOP/s is the clock."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_sound_generation_amd import ops
dev = "cuda:0"
B, D = int(os.environ.get("B", "128")), int(os.environ.get("D", "128"))
DT = torch.bfloat16
for name, (k, s_, ih, iw, tr) in {"3x3 fwd": (3, 1, 20, 256, False), "4x4/s2 fwd": (4, 2, 40, 512, False), "convT fwd": (4, 2, 20, 256, True)}.items():
    d = ops.conv_desc(B, ih, iw, D, D, k, s_, 1, transposed=tr, dtype=DT)
    w = torch.randn(D, D, k, k, device=dev) * 0.05
    wf, _ = ops.pack_weights(d, w)
    wz, _ = ops.pack_weights(d, torch.zeros_like(w))
    bias = torch.zeros(D, device=dev)
    big = torch.randn(B, 40, 512, D, device=dev).to(DT)            # 671 MB: the streaming kernel's tensor
    flops = 2.0 * B * (ih * iw if tr else d.OH * d.OW) * k * k * D * D
    datas = {"zeros x, random w": (torch.zeros(B, ih, iw, D, device=dev).to(DT), wf),
             "relu(randn) x, random w": (torch.relu(torch.randn(B, ih, iw, D, device=dev)).to(DT), wf),
             "randn x, random w": (torch.randn(B, ih, iw, D, device=dev).to(DT), wf),
             "randn x, zero w": (torch.randn(B, ih, iw, D, device=dev).to(DT), wz)}
    for dn, (x, wp) in datas.items():
        for mode in ("back to back", "alternating with a 671 MB convert pass"):
            for _ in range(5):
                ops.conv_forward(d, x, wp, bias)
            torch.cuda.synchronize()
            ts = []
            for _ in range(20):
                if mode != "back to back":
                    ops.convert(big, torch.bfloat16, out=big, relu=False)
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); ops.conv_forward(d, x, wp, bias); b.record()
                torch.cuda.synchronize()
                ts.append(a.elapsed_time(b) * 1e3)
            us = float(np.median(ts[5:]))
            print(f"{name:11s} {dn:26s} {mode:40s} {us:7.1f} us = {flops / us / 1e6:6.0f} TFLOP/s")
