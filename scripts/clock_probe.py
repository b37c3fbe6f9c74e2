#!/usr/bin/env python
"""Diagnostics: (1) fp32-MFMA rate and clock of a register-only loop with every CU busy,
(2) in-kernel clock and per-block cycles of gather_gemm_f32 on the 3x3 ResBlock conv at bench shape."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_sound_generation_amd import _lib, ops
lib = _lib.use_diag().__enter__()      # the diagnostics library (libnsg_diag.so: switches, stamps, probe kernels) for this whole process
lib.nsg_debug_mfma_peak.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
lib.nsg_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p]
dev = "cuda:0"
for blocks_per_cu in (1, 2):
    blocks, iters = 256 * blocks_per_cu, 20000
    sink = torch.empty(blocks * 256, device=dev); stamps = torch.zeros(blocks * 2, dtype=torch.int64, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        lib.nsg_debug_mfma_peak(blocks, iters, sink.data_ptr(), stamps.data_ptr(), st)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        lib.nsg_debug_mfma_peak(blocks, iters, sink.data_ptr(), stamps.data_ptr(), st)
    b.record(); torch.cuda.synchronize()
    sec = a.elapsed_time(b) / 5 * 1e-3
    flops = blocks * 4 * iters * 16 * 2.0 * 32 * 32 * 2
    s = stamps.cpu().numpy().reshape(-1, 2).astype(np.float64)
    clk = np.median(s[:, 0] / s[:, 1]) * 100e6
    print(f"pure MFMA loop, {blocks_per_cu} block(s)/CU x 4 waves: {flops / sec / 1e12:6.1f} TFLOP/s, in-kernel clock {clk / 1e9:.3f} GHz, "
          f"cycles per MFMA per SIMD {np.median(s[:, 0]) / (iters * 16) / blocks_per_cu * 1:.1f} (x{blocks_per_cu} waves)")
B, D = 64, 128
for name, (k, s_, p_, tr, ih, iw) in {"3x3": (3, 1, 1, False, 20, 256), "4x4s2": (4, 2, 1, False, 40, 512), "1x1": (1, 1, 0, False, 20, 256)}.items():
    d = ops.conv_desc(B, ih, iw, D, D, k, s_, p_, transposed=tr)
    x = torch.randn(B, ih, iw, D, device=dev); w = torch.randn(D, D, k, k, device=dev) * 0.05
    wf, _ = ops.pack_weights(d, w); bias = torch.zeros(D, device=dev)
    nblk = (B * d.OH * d.OW + 127) // 128
    stamps = torch.zeros(nblk * 4, dtype=torch.int64, device=dev)
    for _ in range(3):
        ops.conv_forward(d, x, wf, bias, flags=ops.NSG_RELU_IN)
    lib.nsg_debug_set_stamp_buffer(stamps.data_ptr())
    ops.conv_forward(d, x, wf, bias, flags=ops.NSG_RELU_IN)
    torch.cuda.synchronize()
    lib.nsg_debug_set_stamp_buffer(None)
    s = stamps.cpu().numpy().reshape(-1, 4).astype(np.float64)
    clk = float('nan')   # (gather_gemm's realtime stamp slot now carries the epilogue split: see scripts/bf16_loop_share.py)
    nit = k * k * (D // 32)
    ideal = nit * 64 * 64          # cycles of MFMA work per wave (64 MFMAs x 64 cycles per chunk)
    print(f"gather_gemm {name}: in-kernel clock {clk / 1e9:.3f} GHz; main loop {np.median(s[:, 0]):.0f} cycles/block median "
          f"(min {s[:, 0].min():.0f}, max {s[:, 0].max():.0f}); MFMA-only would be {ideal} per wave, x2 waves/SIMD = {2 * ideal}")

lib.nsg_debug_set_wgrad_stamp_buffer.argtypes = [ctypes.c_void_p]
for name, (k, s_, p_, tr, ih, iw) in {"3x3": (3, 1, 1, False, 20, 256), "4x4s2": (4, 2, 1, False, 40, 512), "1x1": (1, 1, 0, False, 20, 256)}.items():
    d = ops.conv_desc(B, ih, iw, D, D, k, s_, p_, transposed=tr)
    x = torch.randn(B, ih, iw, D, device=dev); dy = torch.randn(B, d.OH, d.OW, D, device=dev)
    for _ in range(3):
        ops.conv_wgrad(d, x, dy, (D, D, k, k), want_bias=False)
    stamps = torch.zeros(2 * 4096, dtype=torch.int64, device=dev)
    lib.nsg_debug_set_wgrad_stamp_buffer(stamps.data_ptr())
    ops.conv_wgrad(d, x, dy, (D, D, k, k), want_bias=False)
    torch.cuda.synchronize()
    lib.nsg_debug_set_wgrad_stamp_buffer(None)
    s = stamps.cpu().numpy().reshape(-1, 2).astype(np.float64)
    s = s[s[:, 1] > 0]
    clk = np.median(s[:, 0] / s[:, 1]) * 100e6
    rows = B * d.OH * d.OW
    nblk = len(s)
    chunks = rows / (nblk / (k * k)) / 32
    ideal = chunks * 64 * 64
    print(f"wgrad_gemm {name}: {nblk} blocks, in-kernel clock {clk / 1e9:.3f} GHz; main loop {np.median(s[:, 0]):.0f} cycles/block median "
          f"(min {s[:, 0].min():.0f}, max {s[:, 0].max():.0f}); MFMA-only {ideal:.0f} per wave, x2 waves/SIMD = {2 * ideal:.0f}")
