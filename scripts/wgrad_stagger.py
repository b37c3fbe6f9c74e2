import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_sound_generation_amd import _lib, ops
lib = _lib.use_diag().__enter__()      # the diagnostics library (libnsg_diag.so: switches, stamps, probe kernels) for this whole process
lib.nsg_debug_set_wgrad_stagger.argtypes = [ctypes.c_int]
dev="cuda:0"; B,D=64,128
def timeit(fn, it=10):
    fn(); fn(); torch.cuda.synchronize()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/it*1e3
for name,(k,s_,p_,ih,iw) in {"3x3":(3,1,1,20,256),"4x4s2":(4,2,1,40,512),"1x1":(1,1,0,20,256)}.items():
    d = ops.conv_desc(B, ih, iw, D, D, k, s_, p_)
    x = torch.randn(B, ih, iw, D, device=dev); dy = torch.randn(B, d.OH, d.OW, D, device=dev)
    dw = torch.empty(D,D,k,k,device=dev)
    res=[]
    for st in (0, 16, 32, 48, 64, 96, 128, 192):
        lib.nsg_debug_set_wgrad_stagger(st)
        res.append((st, round(timeit(lambda: ops.conv_wgrad(d, x, dy, (D,D,k,k), dw=dw, want_bias=False)),1)))
    lib.nsg_debug_set_wgrad_stagger(0)
    print(name, res)
