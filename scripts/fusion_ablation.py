"""Which fused operator moves the bf16 step's gradients how far from the separate-operator step (same seed, same batch):
gradient cosines per switch.  Usage: python scripts/fusion_ablation.py"""
import sys
import torch
sys.path.insert(0, '.')
from neural_sound_generation_amd import engine, models as M, train as T
from neural_sound_generation_amd.train import FusedTrainStep

DEV = "cuda:0"
c = torch.rand(4, 1, 80, 512, generator=torch.Generator().manual_seed(77)).to(DEV)
SW = {"FUSED_C1_LAYER": engine, "FUSED_OUT_LAYER": engine, "FUSED_1X1": engine, "LEAN_VQ": T}


def run(on):
    for k, mod in SW.items():
        setattr(mod, k, k in on)
    torch.manual_seed(3)
    m = M.VQVAE(1, 128, 512, compute_dtype=torch.bfloat16).to(DEV).train()
    st = FusedTrainStep(m, lr=1e-3)
    l = st.forward_backward(c)
    return [float(x) for x in l], st.last_indices.clone(), {k: p.grad.detach().clone() for k, p in m.named_parameters()}


base = run(())
for on in [(), ("FUSED_C1_LAYER",), ("FUSED_OUT_LAYER",), ("FUSED_1X1",), ("LEAN_VQ",), tuple(SW)]:
    l, idx, g = run(on)
    cos = {}
    for k in g:
        a, b = g[k].double().flatten(), base[2][k].double().flatten()
        if b.norm() > 1e-7:
            cos[k] = float(torch.dot(a, b) / (a.norm() * b.norm()))
    v = sorted(cos.values())
    worst = sorted(cos.items(), key=lambda kv: kv[1])[:4]
    print(on, "losses", [round(x, 6) for x in l[:2]], "flips %.4f" % float((idx != base[1]).float().mean()), "min %.4f median %.4f" % (v[0], v[len(v) // 2]),
          [(k, round(x, 3)) for k, x in worst])
