#!/usr/bin/env python
"""Per-kernel micro-benchmark at the bench shapes (B clips of 80x1024, D channels, K codes):
HIP-event time and achieved TFLOP/s or GB/s for each layer's forward / dgrad / wgrad and the VQ /
BatchNorm kernels.  Development tool; bench.py is the contract benchmark."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_sound_generation_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--dim", type=int, default=128)
ap.add_argument("--z-dim", type=int, default=512)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--only", type=str, default="")
ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32")
args = ap.parse_args()
B, D, K, IT = args.batch, args.dim, args.z_dim, args.iters
DT = torch.bfloat16 if args.dtype == "bf16" else torch.float32
PEAK = 2500.0 if args.dtype == "bf16" else 157.3
dev = "cuda:0"


def timeit(fn):
    fn(); fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(IT):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / IT * 1e-3


def report(name, sec, flops=None, bytes_=None):
    s = f"{name:44s} {sec * 1e6:9.1f} us"
    if flops:
        s += f"  {flops / sec / 1e12:7.1f} TFLOP/s ({flops / sec / 1e12 / PEAK * 100:4.1f}% {args.dtype}-MFMA)"
    if bytes_:
        s += f"  {bytes_ / sec / 1e9:8.0f} GB/s"
    print(s, flush=True)


LAYERS = [
    ("enc0  conv4x4s2 1->D   @80x1024", (B, 80, 1024, 1, D, 4, 2, 1, False)),
    ("enc3  conv4x4s2 D->D   @40x512", (B, 40, 512, D, D, 4, 2, 1, False)),
    ("res   conv3x3   D->D   @20x256", (B, 20, 256, D, D, 3, 1, 1, False)),
    ("res   conv1x1   D->D   @20x256", (B, 20, 256, D, D, 1, 1, 0, False)),
    ("dec3  convT4x4s2 D->D  @20x256", (B, 20, 256, D, D, 4, 2, 1, True)),
    ("dec6  convT4x4s2 D->1  @40x512", (B, 40, 512, D, 1, 4, 2, 1, True)),
]
for name, (b, ih, iw, ci, co, k, s, p, tr) in LAYERS:
    if args.only and args.only not in name:
        continue
    d = ops.conv_desc(b, ih, iw, ci, co, k, s, p, transposed=tr, dtype=DT)
    x = torch.randn(b, ih, iw, ci, device=dev).to(DT if ci > 1 else torch.float32)
    dy = torch.randn(b, d.OH, d.OW, co, device=dev).to(DT if co > 1 else torch.float32)
    wshape = (ci, co, k, k) if tr else (co, ci, k, k)
    w = torch.randn(*wshape, device=dev) * 0.05
    bias = torch.zeros(co, device=dev)
    wf, wd = ops.pack_weights(d, w)
    fl = ops._gemm_flops(d)
    y = torch.empty(b, d.OH, d.OW, co, device=dev, dtype=DT if co > 1 else torch.float32)
    dx = torch.empty(b, ih, iw, ci, device=dev, dtype=DT if ci > 1 else torch.float32)
    dw = torch.empty(wshape, device=dev)
    db = torch.empty(co, device=dev)
    io_bytes = float(x.numel() * x.element_size() + y.numel() * y.element_size())
    report(name + " fwd", timeit(lambda: ops.conv_forward(d, x, wf, bias, out=y)), fl, io_bytes)
    report(name + " dgrad", timeit(lambda: ops.conv_dgrad(d, dy, wd, out=dx)), fl, io_bytes)
    report(name + " wgrad(+bias)", timeit(lambda: ops.conv_wgrad(d, x, dy, wshape, dw=dw, dbias=db)), fl, io_bytes)

if not args.only or "vq" in args.only:
    N = B * 20 * 256
    z = torch.randn(N, D, device=dev)
    e = (torch.rand(K, D, device=dev) * 2 - 1) / K
    report(f"vq_forward N={N} K={K} D={D}", timeit(lambda: ops.vq_forward(z, e)), 2.0 * N * K * D, 4.0 * 2 * N * D)
    idx, _, _ = ops.vq_forward(z, e)
    report("index_add_rows (codebook grad)", timeit(lambda: ops.index_add_rows(idx, z, K)), 2.0 * N * K * D, 4.0 * N * D)

if not args.only or "bn" in args.only:
    for hh, ww in ((20, 256), (40, 512)):
        M = B * hh * ww
        x = torch.randn(M, D, device=dev).to(DT)
        dy = torch.randn(M, D, device=dev).to(DT)
        g = torch.ones(D, device=dev)
        bt = torch.zeros(D, device=dev)
        nb = float(x.element_size()) * M * D
        mean, invstd = ops.bn_stats(x, D)
        y = torch.empty_like(x)
        dx = torch.empty_like(x)
        report(f"bn_stats      M={M}", timeit(lambda: ops.bn_stats(x, D)), None, 2 * nb)
        report(f"bn_apply+relu M={M}", timeit(lambda: ops.bn_apply(x, mean, invstd, g, bt, relu=True, out=y)), None, 2 * nb)
        report(f"bn_backward   M={M}", timeit(lambda: ops.bn_backward(x, y, dy, mean, invstd, g, out=dx)), None, 7 * nb)
        report(f"relu_bwd_add  M={M}", timeit(lambda: ops.relu_backward_add(dy, x, y, out=dx)), None, 4 * nb)
