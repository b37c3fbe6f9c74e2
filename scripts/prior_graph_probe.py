"""Is the latent prior's training step launch-bound?  Eager autograd step vs the same step replayed from a HIP graph
(torch.optim.Adam(capturable=True)); GatedPixelCNN(512, 64, 15) on B x 20 x 256 codes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_sound_generation_amd.prior import GatedPixelCNN

dev = torch.device("cuda:0")
for B in (16, 64):
    torch.manual_seed(1)
    m = GatedPixelCNN(512, 64, 15, 10).to(dev)
    opt = torch.optim.Adam(m.parameters(), lr=3e-4, capturable=True)
    x = torch.randint(0, 512, (B, 20, 256), device=dev)
    y = torch.randint(0, 10, (B,), device=dev)

    def step():
        opt.zero_grad(set_to_none=False)
        l = m.loss(x, y)
        l.backward()
        opt.step()
        return l

    def timeit(fn, n=10):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n * 1e3

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(s)
    eager = timeit(step)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        loss = step()
    replay = timeit(g.replay)
    print(f"B={B}: eager {eager:.2f} ms/step, graph replay {replay:.2f} ms/step, loss {loss.item():.4f}")
