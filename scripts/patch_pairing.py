"""Which workgroups of a 512-workgroup patch_gemm launch share a CU (diagnostics build: HW_ID / XCC_ID in the stamp record)."""
import os, sys, collections
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_sound_generation_amd import _lib, ops
lib = _lib.use_diag().__enter__()
dev = "cuda:0"
B, D = 128, 128
d = ops.conv_desc(B, 20, 256, D, D, 3, 1, 1, dtype=torch.bfloat16)
x = torch.relu(torch.randn(B, 20, 256, D, device=dev)).to(torch.bfloat16)
wf, _ = ops.pack_weights(d, torch.randn(D, D, 3, 3, device=dev) * 0.05)
bias = torch.zeros(D, device=dev)
for trial in range(3):
    stamps = torch.zeros(65536 * 8, dtype=torch.int64, device=dev)
    lib.nsg_debug_set_stamp_buffer(stamps.data_ptr())
    ops.conv_forward(d, x, wf, bias); torch.cuda.synchronize()
    lib.nsg_debug_set_stamp_buffer(None)
    s = stamps.cpu().numpy().reshape(-1, 8)[:512]
    ids = s[:, 7]
    xcc = (ids >> 32) & 0xf
    hw = ids & 0xffffffff
    cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
    key = [(int(a), int(b), int(c), int(e)) for a, b, c, e in zip(xcc, se, sh, cu)]
    groups = collections.defaultdict(list)
    for i, k in enumerate(key):
        groups[k].append(i)
    sizes = collections.Counter(len(v) for v in groups.values())
    diffs = collections.Counter((v[1] - v[0]) for v in groups.values() if len(v) == 2)
    print(f"trial {trial}: {len(groups)} distinct CUs, workgroups per CU {dict(sizes)}, index distance of the two sharing a CU: {dict(diffs.most_common(6))}")
    print("   blockIdx.x % 8 -> XCC of the first 16 workgroups:", [int(v) for v in xcc[:16]])
