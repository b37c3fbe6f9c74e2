"""When do the 512 workgroups of a patch_gemm launch start and end (diagnostics build with -DNSG_PATCH_ABS_STAMPS)?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_sound_generation_amd import _lib, ops
lib = _lib.use_diag().__enter__()
dev = "cuda:0"
B, D = 128, 128
for name, (k, s_, ih, iw, tr) in {"3x3 fwd": (3, 1, 20, 256, False), "4x4/s2 fwd": (4, 2, 40, 512, False), "convT fwd": (4, 2, 20, 256, True)}.items():
    d = ops.conv_desc(B, ih, iw, D, D, k, s_, 1, transposed=tr, dtype=torch.bfloat16)
    x = torch.relu(torch.randn(B, ih, iw, D, device=dev)).to(torch.bfloat16)
    wf, _ = ops.pack_weights(d, torch.randn(D, D, k, k, device=dev) * 0.05)
    bias = torch.zeros(D, device=dev)
    for _ in range(5):
        ops.conv_forward(d, x, wf, bias)
    torch.cuda.synchronize()
    stamps = torch.zeros(65536 * 8, dtype=torch.int64, device=dev)
    lib.nsg_debug_set_stamp_buffer(stamps.data_ptr())
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ops.conv_forward(d, x, wf, bias)          # (queue something in front so the timed launch is not host-bound)
    a.record(); ops.conv_forward(d, x, wf, bias); b.record(); torch.cuda.synchronize()
    lib.nsg_debug_set_stamp_buffer(None)
    s = stamps.cpu().numpy().reshape(-1, 8)[:512].astype(np.float64)
    st, en = s[:, 3] / 100.0, s[:, 5] / 100.0          # us
    t0 = st.min()
    life = en - st
    pct = lambda v: [round(float(np.percentile(v, q)), 1) for q in (0, 10, 50, 90, 100)]
    print(f"{name}: event time {a.elapsed_time(b) * 1e3:.1f} us; first start -> last end {en.max() - t0:.1f} us; starts (0/10/50/90/100 %) {pct(st - t0)}; "
          f"ends {pct(en - t0)}; lives {pct(life)}; first half lives {pct(life[:256])}, second half {pct(life[256:])}")
