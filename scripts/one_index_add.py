"""index_add_rows at a training-step shape, a few times (for rocprofv3 --kernel-trace --stats): python scripts/one_index_add.py N D K impl"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_sound_generation_amd import ops
N, D, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
impl = sys.argv[4] if len(sys.argv) > 4 else "sorted"
g = torch.Generator().manual_seed(0)
# spatially correlated codes, as a trained quantiser gives: runs of equal codes
runs = torch.randint(0, K, (N // 6 + 1,), generator=g).repeat_interleave(6)[:N]
idx = runs.to("cuda:0")
v = torch.randn(N, D, device="cuda:0")
for _ in range(5):
    out, cnt = ops.index_add_rows(idx, v, K, want_counts=True, impl=impl)
torch.cuda.synchronize()
print("ok", float(out.abs().mean()))
