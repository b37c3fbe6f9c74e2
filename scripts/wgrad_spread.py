import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from neural_sound_generation_amd import _lib, ops
lib = _lib.use_diag().__enter__()      # the diagnostics library (libnsg_diag.so: switches, stamps, probe kernels) for this whole process
lib.nsg_debug_set_wgrad_stamp_buffer.argtypes = [ctypes.c_void_p]
dev="cuda:0"; B,D=64,128
k,s_,p_,ih,iw=3,1,1,20,256
d = ops.conv_desc(B, ih, iw, D, D, k, s_, p_)
x = torch.randn(B, ih, iw, D, device=dev); dy = torch.randn(B, d.OH, d.OW, D, device=dev)
for _ in range(3): ops.conv_wgrad(d, x, dy, (D,D,k,k), want_bias=False)
stamps = torch.zeros(2*4096, dtype=torch.int64, device=dev)
lib.nsg_debug_set_wgrad_stamp_buffer(stamps.data_ptr())
ops.conv_wgrad(d, x, dy, (D,D,k,k), want_bias=False); torch.cuda.synchronize()
lib.nsg_debug_set_wgrad_stamp_buffer(None)
s = stamps.cpu().numpy().reshape(-1,2).astype(np.float64); n=(s[:,1]>0).sum(); s=s[:n]
nslab=n//9
cyc=s[:,0].reshape(9,nslab)   # bid = tap*nslab + slab
print("nslab",nslab,"percentiles", np.percentile(s[:,0],[0,10,25,50,75,90,100]).astype(int))
print("per-tap median", np.median(cyc,axis=1).astype(int))
print("per-slab median (first 16)", np.median(cyc,axis=0).astype(int)[:16])
print("per-(slab%8) median", [int(np.median(cyc[:, i::8])) for i in range(8)])
rt=s[:,1].reshape(9,nslab)
print("realtime ticks per-tap median", np.median(rt,axis=1).astype(int))
