import sys, time, torch
sys.path.insert(0, '.')
from neural_sound_generation_amd import ops
dev = 'cuda:0'
for (N, D, K) in ((655360, 128, 512), (81920, 256, 8192), (655360, 256, 8192)):
    x = torch.randn(N, D, device=dev); e = torch.randn(K, D, device=dev) * 0.5
    for impl in ("bf16x3", "mfma"):
        kw = dict(want_codes=False, impl=impl)
        for _ in range(2): ops.vq_forward(x, e, **kw)
        torch.cuda.synchronize(); t = time.perf_counter()
        n = 5
        for _ in range(n): ops.vq_forward(x, e, **kw)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
        fl = 2.0 * N * D * K
        print(f"N={N} D={D} K={K} {impl:7s} {dt*1e3:8.3f} ms  {fl/dt/1e12:7.1f} TF algorithmic ({3*fl/dt/1e12 if impl=='bf16x3' else fl/dt/1e12:7.1f} TF issued)")
