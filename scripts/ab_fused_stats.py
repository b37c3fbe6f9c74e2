import os, sys, torch, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from neural_sound_generation_amd import models as M, engine
from neural_sound_generation_amd.train import FusedTrainStep
from oracle import vqvae_oracle as O
torch.set_num_threads(16)
for dim, z in ((64,128),(128,512)):
    torch.manual_seed(1); model0 = M.VQVAE(1, dim, z); st0 = O.clone_state(model0.state_dict())
    c = torch.rand(2,1,80,1024, generator=torch.Generator().manual_seed(1234))
    rec = O.forward_backward(st0, c)
    st64 = O.clone_state({k: (v.double() if v.is_floating_point() else v) for k, v in st0.items()})
    rec64 = O.forward_backward(st64, c.double())
    skip = ("encoder.0.bias", "block.1.bias", "block.4.bias", "decoder.3.bias")
    e32 = {k: float((rec["grads"][k].double() - rec64["grads"][k]).norm() / max(rec64["grads"][k].norm(), 1e-12)) for k in rec["grads"] if not k.endswith(skip)}
    print(dim, z, "fp64 idx equal:", bool((rec64["idx"] == rec["idx"]).all()), " CPU-fp32 oracle vs fp64 truth, worst:", [(k, "%.1e" % v) for k, v in sorted(e32.items(), key=lambda kv: -kv[1])[:4]])
    res = {}
    for fused in (False, True):
        engine.FUSED_BN_STATS = fused
        m = M.VQVAE(1, dim, z); m.load_state_dict(st0); m = m.cuda().train()
        st = FusedTrainStep(m, lr=1e-3); l = st.forward_backward(c.cuda())
        flips = int((st.last_indices.cpu().numpy() != rec["idx"].numpy()).sum())
        errs = {}
        for k, p in m.named_parameters():
            if k.endswith(("encoder.0.bias", "block.1.bias", "block.4.bias", "decoder.3.bias")):
                continue  # exact-zero gradients (bias in front of a BatchNorm): pure round-off on both sides
            g, r = p.grad.double().cpu(), rec["grads"][k].double()
            errs[k] = float((g - r).norm() / max(r.norm(), 1e-12))
        res[fused] = (l[0].item(), l[1].item(), flips, errs, {k: p.grad.double().cpu().clone() for k, p in m.named_parameters()})
    print(dim, z, "oracle losses", rec["loss_recons"].item(), rec["loss_vq"].item())
    for fused in (False, True):
        l0, l1, flips, errs, _ = res[fused]
        worst = sorted(errs.items(), key=lambda kv: -kv[1])[:6]
        print(" fused" if fused else " plain", "loss relerr %.2e %.2e flips %d" % (abs(l0-rec["loss_recons"].item())/rec["loss_recons"].item(), abs(l1-rec["loss_vq"].item())/rec["loss_vq"].item(), flips))
        for k, v in worst: print("      rel L2 err vs fp32 oracle %-34s %.2e" % (k, v))
        e64 = {k: float((res[fused][4][k] - rec64["grads"][k]).norm() / max(rec64["grads"][k].norm(), 1e-12)) for k in errs}
        print("      vs fp64 truth, worst:", [(k, "%.1e" % v) for k, v in sorted(e64.items(), key=lambda kv: -kv[1])[:4]])
    d = {k: float((res[True][4][k]-res[False][4][k]).norm()/max(res[False][4][k].norm(),1e-12)) for k in res[True][3]}
    print("  fused vs plain (GPU vs GPU) worst:", sorted(d.items(), key=lambda kv: -kv[1])[:4])
