"""bf16 gather_gemm: how much of a workgroup's life is the main loop?  (stamps: cycles spent between the first staging
load and the last MFMA; kernel time from HIP events.)"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_sound_generation_amd import _lib, ops
lib = _lib.use_diag().__enter__()      # the diagnostics library (libnsg_diag.so: switches, stamps, probe kernels) for this whole process
lib.nsg_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p]
dev = "cuda:0"
B, D = 64, 128
DT = torch.bfloat16
for name, (k, s_, p_, ih, iw) in {"3x3": (3, 1, 1, 20, 256), "4x4s2": (4, 2, 1, 40, 512), "1x1": (1, 1, 0, 20, 256)}.items():
    d = ops.conv_desc(B, ih, iw, D, D, k, s_, p_, dtype=DT)
    x = torch.relu(torch.randn(B, ih, iw, D, device=dev)).to(DT)
    w = torch.randn(D, D, k, k, device=dev) * 0.05
    wf, _ = ops.pack_weights(d, w); bias = torch.zeros(D, device=dev)
    nblk = (B * d.OH * d.OW + 127) // 128
    stamps = torch.zeros(nblk * 4, dtype=torch.int64, device=dev)
    y = torch.empty(B, d.OH, d.OW, D, device=dev, dtype=DT)
    for _ in range(3):
        ops.conv_forward(d, x, wf, bias, out=y)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    lib.nsg_debug_set_stamp_buffer(stamps.data_ptr())
    a.record()
    ops.conv_forward(d, x, wf, bias, out=y)
    b.record()
    torch.cuda.synchronize()
    lib.nsg_debug_set_stamp_buffer(None)
    s = stamps.cpu().numpy().reshape(-1, 4).astype(np.float64)
    clk = 1.9e9   # (the realtime stamp slot now carries the epilogue split)
    t = a.elapsed_time(b) * 1e-3
    rounds = nblk / 512.0
    per_block_total = t / rounds * clk            # cycles a resident workgroup lives if the kernel were perfectly round-structured
    nit = k * k * (D // 64)
    print(f"{name}: kernel {t * 1e6:.1f} us, clock {clk / 1e9:.2f} GHz, {nblk} workgroups = {rounds:.1f} rounds of 512; main loop {np.median(s[:, 0]):.0f} cycles "
          f"median ({nit} chunks -> {np.median(s[:, 0]) / nit:.0f} per chunk; MFMA-only for the SIMD's two waves {2 * nit * 16 * 32}); a workgroup's slot lasts ~{per_block_total:.0f} cycles "
          f"-> main loop share {np.median(s[:, 0]) / per_block_total:.2f}; prologue {np.median(s[:, 2]):.0f}, epilogue {np.median(s[:, 3]):.0f} cycles (of which until the tile stands in LDS: {np.median(s[:, 1]):.0f})")
