import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_sound_generation_amd import ops
dev="cuda:0"
for N, D in ((655360,128),(81920,256)):
    x=torch.randn(N,D,device=dev)
    for _ in range(3): ops.rowsumsq(x)
    torch.cuda.synchronize(); ts=[]
    for _ in range(10):
        a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        a.record(); ops.rowsumsq(x); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b)*1e3)
    us=float(np.median(ts)); print(f"rowsumsq N={N} D={D}: {us:.1f} us = {N*D*4/us/1e6:.2f} TB/s")
