import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_sound_generation_amd import _lib, ops
lib = _lib.load()
lib.nsg_debug_set_wgrad_stamp_buffer.argtypes = [ctypes.c_void_p]
lib.nsg_debug_set_wgrad_diag.argtypes = [ctypes.c_int]
dev="cuda:0"; B,D=64,128
for name,(k,s_,p_,ih,iw) in {"3x3":(3,1,1,20,256),"4x4s2":(4,2,1,40,512)}.items():
    d = ops.conv_desc(B, ih, iw, D, D, k, s_, p_)
    x = torch.randn(B, ih, iw, D, device=dev); dy = torch.randn(B, d.OH, d.OW, D, device=dev)
    for _ in range(3): ops.conv_wgrad(d, x, dy, (D,D,k,k), want_bias=False)
    stamps = torch.zeros(2*4096 + 4096*16, dtype=torch.int64, device=dev)
    lib.nsg_debug_set_wgrad_stamp_buffer(stamps.data_ptr()); lib.nsg_debug_set_wgrad_diag(1)
    ops.conv_wgrad(d, x, dy, (D,D,k,k), want_bias=False); torch.cuda.synchronize()
    lib.nsg_debug_set_wgrad_stamp_buffer(None); lib.nsg_debug_set_wgrad_diag(0)
    s = stamps.cpu().numpy().astype(np.float64)
    tot = s[:2*4096].reshape(-1,2); n=int((tot[:,1]>0).sum())
    seg = s[2*4096:].reshape(-1,4)[:n*4].reshape(n,4,4)   # [block][wave][segment]
    nchunks = (B*d.OH*d.OW) / (n/(k*k)) / 32
    first, second = seg[:256].mean(axis=(0,1))/nchunks, seg[256:n].mean(axis=(0,1))/nchunks
    print(f"{name}: {n} blocks, {nchunks:.0f} chunks/block; cycles per chunk per wave [gload, compute, lstore, barrier]")
    print("   first-dispatched 256 blocks:", first.astype(int), "sum", int(first.sum()))
    print("   later blocks              :", second.astype(int), "sum", int(second.sum()))
