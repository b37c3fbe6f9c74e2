"""One rank of the two-rank data-parallel rehearsal (tests/test_gpu_distributed.py).  Started as a FRESH child process
(before its parent touches the GPU) by tests/conftest.py; both ranks share GPU 0 and talk over gloo
(NSG_DIST_BACKEND=gloo NSG_DEVICE_INDEX=0), so the product's own FusedTrainStep.step() -- HIP forward / backward, the
world > 1 branch with its ONE all-reduce, the Adam kernel -- runs exactly as it would under torchrun + RCCL.

    python tests/helpers/dp_rank.py <golden model_tiny.npz> <out dir>        (RANK / WORLD_SIZE / MASTER_* in the env)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    golden_path, out_dir = sys.argv[1], sys.argv[2]
    from neural_sound_generation_amd import distributed as D, models as M
    from neural_sound_generation_amd.train import FusedTrainStep

    rank, world, _ = D.init_from_env()
    dev = torch.device("cuda", int(os.environ.get("NSG_DEVICE_INDEX", "0")))
    torch.cuda.set_device(dev)
    g = np.load(golden_path)
    out = {}

    # ---- (1) gradient mode against the reference-generated DP fixture (dp.*: R = 2 replicas, per-rank BatchNorm,
    #          averaged gradients, one Adam step).  Only rank 0 holds the fixture's weights and buffers; rank 1 starts from a
    #          different seed with perturbed BatchNorm buffers: FusedTrainStep must replicate rank 0's state itself.
    torch.manual_seed(100 + rank)
    model = M.VQVAE(1, 16, 32)
    if rank == 0:
        model.load_state_dict({k[4:]: torch.from_numpy(np.array(g[k])) for k in g.files if k.startswith("sd0.")})
    else:
        for b in model.buffers():
            if b.is_floating_point():
                b.add_(0.5)
            else:
                b.add_(7)
    model = model.to(dev).train()
    step = FusedTrainStep(model, lr=1e-3, beta=1.0)
    assert step.world == world == 2
    out["after_init." + str(rank)] = {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}
    c = torch.from_numpy(g["dp.c%d" % rank]).to(dev)
    losses = step.step(c)
    torch.cuda.synchronize()
    out["losses"] = np.array([x.item() for x in losses])
    out["grad"] = {k: (p.grad * 0.5).cpu().numpy() for k, p in model.named_parameters()}     # the bucket holds the SUM
    out["sd1"] = {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}

    # ---- (2) EMA codebook mode: per-code counts and sums ride behind the gradients in the same all-reduce
    torch.manual_seed(1)
    ema = M.VQVAE(1, 16, 32, ema_decay=0.99)
    if rank == 1:       # a resumed checkpoint loaded on rank 0 only: rank 1's codebook and EMA buffers differ until the broadcast
        ema.codebook.embedding.weight.data.mul_(3.0)
        ema.codebook.ema_count.add_(1.0)
        ema.codebook.ema_sum.add_(1.0)
    ema = ema.to(dev).train()
    estep = FusedTrainStep(ema, lr=1e-3)
    out["ema_after_init"] = {k: v.detach().cpu().numpy().copy() for k, v in ema.state_dict().items()}
    estep.step(c)
    torch.cuda.synchronize()
    out["ema_n"] = estep.ema_n.cpu().numpy()
    out["ema_s"] = estep.ema_s.cpu().numpy()
    out["ema_codebook"] = ema.codebook.embedding.weight.detach().cpu().numpy()
    out["ema_count"] = ema.codebook.ema_count.cpu().numpy()

    flat = {}
    for k, v in out.items():
        if isinstance(v, dict):
            for kk, vv in v.items():
                flat[k + "/" + kk] = vv
        else:
            flat[k] = v
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), **flat)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
