"""gemm_patch.hip: where a workgroup's cycles go (diagnostics build with s_memtime stamps; read the SHARES, not the
length) and the clock the chip holds, per layer shape of the bf16 training step.  Also times the un-stamped kernel."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_sound_generation_amd import _lib, ops
lib = _lib.use_diag().__enter__()      # the diagnostics library (libnsg_diag.so: switches, stamps, probe kernels) for this whole process
lib.nsg_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p]
dev = "cuda:0"
B, D = int(os.environ.get("B", "128")), int(os.environ.get("D", "128"))
DT = torch.bfloat16
grid = int(os.environ.get("NSG_PATCH_GRID", "0"))
shapes = {"3x3 fwd": (3, 1, 1, 20, 256, False, "f"), "3x3 dgrad+add+mask": (3, 1, 1, 20, 256, False, "d"),
          "4x4/s2 fwd": (4, 2, 1, 40, 512, False, "f"), "convT fwd": (4, 2, 1, 20, 256, True, "f")}
for name, (k, s_, p_, ih, iw, tr, role) in shapes.items():
    d = ops.conv_desc(B, ih, iw, D, D, k, s_, p_, transposed=tr, dtype=DT)
    x = torch.relu(torch.randn(B, ih, iw, D, device=dev)).to(DT)
    w = torch.randn(D, D, k, k, device=dev) * 0.05
    wf, wd = ops.pack_weights(d, w); bias = torch.zeros(D, device=dev)
    dy = torch.randn(B, d.OH, d.OW, D, device=dev).to(DT)
    skip = torch.randn(B, ih, iw, D, device=dev).to(DT)
    def run():
        if role == "f":
            return ops.conv_forward(d, x, wf, bias)
        return ops.conv_dgrad(d, dy, wd, add=skip, relu_x=x)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    flops = 2.0 * B * (ih * iw if tr else d.OH * d.OW) * k * k * D * D
    us = float(np.median(ts))
    nwg = 65536
    stamps = torch.zeros(nwg * 8, dtype=torch.int64, device=dev)
    lib.nsg_debug_set_stamp_buffer(stamps.data_ptr())
    run(); torch.cuda.synchronize()
    lib.nsg_debug_set_stamp_buffer(None)
    s = stamps.cpu().numpy().reshape(-1, 8).astype(np.float64)
    s = s[s[:, 0] > 0]
    tot = s[:, 0]
    clk = np.median(s[:, 0] / np.maximum(s[:, 1], 1)) * 100e6
    jobs = np.median(s[:, 6])
    print(f"{name}: {us:.1f} us = {flops / us / 1e6:.0f} TFLOP/s (un-stamped); stamped build: {len(s)} workgroups, life {np.median(tot):.0f} cycles at {clk / 1e9:.2f} GHz; "
          f"shares: tap loops {np.median(s[:, 2] / tot):.2f} ({np.median(s[:, 2]) / jobs:.0f} cycles per job), job boundaries {np.median(s[:, 3] / tot):.2f} "
          f"({np.median(s[:, 3]) / jobs:.0f} per job), epilogues {np.median(s[:, 4] / tot):.2f}, prologue {np.median(s[:, 5] / tot):.2f}, "
          f"rest {1 - np.median((s[:, 2] + s[:, 3] + s[:, 4] + s[:, 5]) / tot):.2f}")
