#!/usr/bin/env python
"""Where the bf16 mode's gradient error comes from.  Runs the bf16 training step with the fp32
oracle's code indices forced in (FusedTrainStep.force_indices), so that no difference is due to flipped codes, and prints
per-tensor cosine / relative L2 against the oracle's fp32 gradients, for a few switch settings.

    python scripts/bf16_grad_fidelity.py [dim z_dim B T]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neural_sound_generation_amd import engine, models as M, train as T_  # noqa: E402
from neural_sound_generation_amd.train import FusedTrainStep  # noqa: E402
from oracle import vqvae_oracle as O  # noqa: E402

dim, z_dim, B, T = (int(v) for v in sys.argv[1:5]) if len(sys.argv) >= 5 else (128, 512, 2, 1024)
torch.manual_seed(1)
model0 = M.VQVAE(1, dim, z_dim)
st0 = O.clone_state(model0.state_dict())
c = torch.rand(B, 1, 80, T, generator=torch.Generator().manual_seed(1234))
torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
rec = O.forward_backward(st0, c)
rec64 = O.forward_backward(O.clone_state({k: (v.double() if v.is_floating_point() else v) for k, v in st0.items()}), c.double())
print("fp32 oracle vs fp64 oracle: idx equal:", bool((rec64["idx"] == rec["idx"]).all()))


def run(label, force, dtype=torch.bfloat16, **sw):
    saved = {}
    for k, v in sw.items():
        mod = T_ if k == "LEAN_VQ" else engine
        saved[k] = getattr(mod, k)
        setattr(mod, k, v)
    try:
        m = M.VQVAE(1, dim, z_dim, compute_dtype=dtype)
        m.load_state_dict(st0)
        m = m.to("cuda:0").train()
        st = FusedTrainStep(m, lr=1e-3)
        if force:
            st.force_indices = rec["idx"].reshape(-1).to("cuda:0")
        l = st.forward_backward(c.to("cuda:0"))
        flips = float((st.last_indices.cpu() != rec["idx"].reshape(-1)).float().mean())
        print(f"\n== {label}: loss_recons {l[0].item():.6f} (fp32 {rec['loss_recons'].item():.6f}), loss_vq {l[1].item():.6f} "
              f"(fp32 {rec['loss_vq'].item():.6f}), indices differing {100 * flips:.2f} %")
        for k, p in m.named_parameters():
            r = rec["grads"][k].double().flatten()
            if r.norm() < 1e-9 or k.endswith(("encoder.0.bias", "block.1.bias", "block.4.bias", "decoder.3.bias")):
                continue
            g = p.grad.double().cpu().flatten()
            t = rec64["grads"][k].flatten()
            cos = float(torch.dot(g, r) / (g.norm() * r.norm()))
            print(f"   {k:34s} cos {cos:8.5f}  relL2 {float((g - r).norm() / r.norm()):9.2e}   (fp32 oracle vs fp64: {float((r - t).norm() / t.norm()):8.1e})")
    finally:
        for k, v in saved.items():
            setattr(T_ if k == "LEAN_VQ" else engine, k, v)


run("bf16, own indices", False)
run("bf16, oracle indices forced", True)
run("bf16, forced, fused 1x1 backward off", True, FUSED_1X1_BWD=False)
run("bf16, forced, fused 1x1 off", True, FUSED_1X1=False)
run("bf16, forced, every fused layer off", True, FUSED_1X1=False, FUSED_C1_LAYER=False, FUSED_OUT_LAYER=False, LEAN_VQ=False)
run("fp32, forced (control)", True, dtype=torch.float32)
