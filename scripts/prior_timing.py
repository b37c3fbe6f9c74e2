"""Timing of one training step of the latent prior (GatedPixelCNN(512, 64, 15 layers)) on the VQ-VAE's code grid
(B clips x 20 x 256 codes), autograd path + torch Adam; reports ms/step and codes/s."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_sound_generation_amd.prior import GatedPixelCNN

dev = "cuda:0"
torch.manual_seed(1)
for B in (16, 64):
    m = GatedPixelCNN(512, 64, 15, 10).to(dev)
    opt = torch.optim.Adam(m.parameters(), lr=3e-4)
    x = torch.randint(0, 512, (B, 20, 256), device=dev)
    y = torch.randint(0, 10, (B,), device=dev)

    def step():
        opt.zero_grad()
        l = m.loss(x, y)
        l.backward()
        opt.step()
        return l
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t = time.perf_counter()
    n = 10
    for _ in range(n):
        l = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / n
    print(f"prior GatedPixelCNN(512, 64, 15): B={B} grid 20x256: {dt * 1e3:.2f} ms/step, {B * 20 * 256 / dt / 1e6:.2f} M codes/s, loss {l.item():.4f}")
