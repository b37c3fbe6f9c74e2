"""The drop-in path (model(c) -> three F.mse_loss -> loss.backward() -> optimizer.step(), src/train.py:104-148) next to
the fused step, same shapes as bench.py."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_sound_generation_amd import models as M
from neural_sound_generation_amd.optim import FlatAdam
from neural_sound_generation_amd.train import FusedTrainStep, vqvae_loss_terms

dev = "cuda:0"
for dt, B in ((torch.bfloat16, 128), (torch.float32, 64)):
    torch.manual_seed(1)
    c = torch.rand(B, 1, 80, 1024, device=dev)
    for name, OptCls in (("FlatAdam", FlatAdam), ("torch.optim.Adam", torch.optim.Adam)):
        m = M.VQVAE(1, 128, 512, compute_dtype=dt).to(dev).train()
        opt = OptCls(m.parameters(), lr=1e-3)

        def step():
            opt.zero_grad()
            xt, ze, zq = m(c)
            lr_, lv, lc = vqvae_loss_terms(c, xt, ze, zq)
            (lr_ + lv + 1.0 * lc).backward()
            opt.step()
        for _ in range(3):
            step()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(10):
            step()
        torch.cuda.synchronize(); d = (time.perf_counter() - t) / 10
        print(f"autograd path, {name:16s} {str(dt):15s} B={B}: {d * 1e3:7.2f} ms/step  {B * 1024 / d / 1e6:.2f} M frames/s")
    m = M.VQVAE(1, 128, 512, compute_dtype=dt).to(dev).train()
    st = FusedTrainStep(m)
    for _ in range(3):
        st.step(c)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10):
        st.step(c)
    torch.cuda.synchronize(); d = (time.perf_counter() - t) / 10
    print(f"fused step                     {str(dt):15s} B={B}: {d * 1e3:7.2f} ms/step  {B * 1024 / d / 1e6:.2f} M frames/s")
