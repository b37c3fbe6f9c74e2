"""Per-tensor gradient accuracy of the fp32 HIP step: relative L2 distance to an fp64 evaluation of the oracle, beside the
fp32 CPU oracle's own distance.  python scripts/grad_accuracy_table.py DIM Z_DIM B [T]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_sound_generation_amd import models as M
from neural_sound_generation_amd.train import FusedTrainStep
from oracle import vqvae_oracle as O

dim, z_dim, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
T = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
torch.manual_seed(1)
model = M.VQVAE(1, dim, z_dim)
st0 = O.clone_state(model.state_dict())
c = torch.rand(B, 1, 80, T, generator=torch.Generator().manual_seed(1234))
torch.set_num_threads(16)
rec = O.forward_backward(st0, c)
rec64 = O.forward_backward(O.clone_state({k: (v.double() if v.is_floating_point() else v) for k, v in st0.items()}), c.double())
model = model.to("cuda:0").train()
step = FusedTrainStep(model, lr=1e-3)
l = step.forward_backward(c.to("cuda:0"))
print("flips vs fp32 oracle", int((step.last_indices.cpu() != rec["idx"]).sum()), "fp32 vs fp64 oracle", int((rec64["idx"] != rec["idx"]).sum()))
for k, p in model.named_parameters():
    truth = rec64["grads"][k]
    tn = max(truth.norm().item(), 1e-12)
    eg = (p.grad.double().cpu() - truth).norm().item() / tn
    ec = (rec["grads"][k].double() - truth).norm().item() / tn
    print(f"{k:40s} |g|={tn:.3e} gpu {eg:.2e} cpu {ec:.2e} ratio {eg / max(ec, 1e-30):.1f}")
