"""A/B of the 8-wave 256 x 128 gather_gemm tile (nsg_debug_set_gather_dma) on the benchmark layers, bf16, post-ReLU-like data."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_sound_generation_amd import _lib, ops

lib = _lib.load()
lib.nsg_debug_set_gather_dma.argtypes = [ctypes.c_int]
B, D = 64, 128
DT = torch.bfloat16
dev = "cuda:0"


def timeit(fn, it=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e-3


LAYERS = [("enc3 4x4s2", (B, 40, 512, D, D, 4, 2, 1, False)), ("res 3x3", (B, 20, 256, D, D, 3, 1, 1, False)),
          ("res 1x1", (B, 20, 256, D, D, 1, 1, 0, False)), ("dec3 convT", (B, 20, 256, D, D, 4, 2, 1, True))]
for name, (b, ih, iw, ci, co, k, s, p, tr) in LAYERS:
    d = ops.conv_desc(b, ih, iw, ci, co, k, s, p, transposed=tr, dtype=DT)
    x = torch.relu(torch.randn(b, ih, iw, ci, device=dev)).to(DT)
    dy = (torch.randn(b, d.OH, d.OW, co, device=dev) * 0.01).to(DT)
    w = torch.randn(*((ci, co, k, k) if tr else (co, ci, k, k)), device=dev) * 0.05
    bias = torch.zeros(co, device=dev)
    wf, wd = ops.pack_weights(d, w)
    y = torch.empty(b, d.OH, d.OW, co, device=dev, dtype=DT)
    dx = torch.empty(b, ih, iw, ci, device=dev, dtype=DT)
    fl = ops._gemm_flops(d)
    res = {}
    for big in (0, 1):
        lib.nsg_debug_set_gather_dma(big)
        tf = timeit(lambda: ops.conv_forward(d, x, wf, bias, out=y))
        tb = timeit(lambda: ops.conv_dgrad(d, dy, wd, out=dx))
        res[big] = (y.clone(), dx.clone())
        print(f"{name:12s} tile={'LDS-DMA   ' if big else 'reg-staged'}  fwd {tf*1e6:7.1f} us {fl/tf/1e12:7.1f} TF   dgrad {tb*1e6:7.1f} us {fl/tb/1e12:7.1f} TF", flush=True)
    print("   identical outputs:", torch.equal(res[0][0], res[1][0]), torch.equal(res[0][1], res[1][1]))
