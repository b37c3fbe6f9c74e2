"""Generates tests/golden/*.npz by IMPORTING the reference's own hot-path modules.

Runs only in the build container (needs /root/reference); the GPU box uses the committed .npz.
    python tests/golden/make_golden.py

Imported from the reference: src/models.py (VQVAE, ResBlock, VQEmbedding) and
src/vector_quantization.py (vq, vq_st).  src/train.py cannot be imported (librosa/lws/... absent),
so the step below drives the reference model the way train.py:109-136 does: zero_grad, forward,
zero-pad to the input width, three F.mse_loss, backward, torch.optim.Adam(lr=1e-3).step().
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference/src")

import models as ref_models  # noqa: E402
from vector_quantization import vq as ref_vq, vq_st as ref_vq_st  # noqa: E402
import portable_rng  # noqa: E402

torch.set_num_threads(8)


def sd_np(model, prefix="sd."):
    return {prefix + k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}


def ref_step(model, opt, c, beta=1.0):
    """The reference's train_vqvae body for one batch (train.py:109-136)."""
    model.train()
    opt.zero_grad()
    x_tilde, z_e_x, z_q_x = model(c)
    target = torch.zeros(c.size(0), c.size(1), c.size(2), c.size(3))
    target[:, :, :, :x_tilde.size(3)] = x_tilde
    loss_recons = F.mse_loss(target, c)
    loss_vq = F.mse_loss(z_q_x, z_e_x.detach())
    loss_commit = F.mse_loss(z_e_x, z_q_x.detach())
    loss = loss_recons + loss_vq + beta * loss_commit
    loss.backward()
    grads = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
    opt.step()
    return dict(x_tilde=x_tilde.detach(), z_e=z_e_x.detach(), z_q=z_q_x.detach(),
                loss_recons=loss_recons.detach(), loss_vq=loss_vq.detach(), loss_commit=loss_commit.detach(),
                grads=grads)


def fp64_gap(x, e, idx):
    """fp64 distance gap between the best and second-best code per row (fragility indicator)."""
    xd, ed = x.astype(np.float64), e.astype(np.float64)
    d = (xd * xd).sum(1, keepdims=True) + (ed * ed).sum(1)[None, :] - 2.0 * xd @ ed.T
    part = np.partition(d, 1, axis=1)
    return (part[:, 1] - part[:, 0]).astype(np.float64), d.argmin(1)


def make_vq_fixtures():
    out = {}
    # F1: robust, unit-scale codebook
    g = torch.Generator().manual_seed(11)
    x = torch.randn(640, 64, generator=g)
    e = torch.randn(128, 64, generator=g)
    idx = ref_vq(x, e)
    codes, idx_flat = ref_vq_st(x, e)
    assert torch.equal(idx, idx_flat)
    out.update({"f1.x": x.numpy(), "f1.e": e.numpy(), "f1.idx": idx.numpy(), "f1.codes": codes.numpy()})
    # F1b: leading dims + ties: duplicate codebook rows -> first index must win
    x = torch.randn(2, 5, 6, 16, generator=g)
    e = torch.randn(24, 16, generator=g)
    e[13] = e[4]
    e[20] = e[4]
    e[7] = e[2]
    out.update({"f1b.x": x.numpy(), "f1b.e": e.numpy(), "f1b.idx": ref_vq(x, e).numpy()})
    # F1c: all-equal distances (zero codebook) -> index 0 everywhere
    x = torch.randn(70, 32, generator=g)
    e = torch.zeros(40, 32)
    out.update({"f1c.x": x.numpy(), "f1c.e": e.numpy(), "f1c.idx": ref_vq(x, e).numpy()})
    # F2: fragile regime (default-init codebook scale; argmin decided in the last ulps)
    for tag, (N, D, K, seed) in {"f2a": (2048, 64, 128, 101), "f2b": (2048, 128, 512, 102),
                                 "f2c": (256, 256, 8192, 103), "f2d": (1000, 16, 32, 104)}.items():
        xn, en = portable_rng.vq_case(N, D, K, seed)
        xt, et = torch.from_numpy(xn), torch.from_numpy(en)
        idx = ref_vq(xt, et).numpy()
        c2 = torch.sum(et ** 2, dim=1)
        x2 = torch.sum(xt ** 2, dim=1, keepdim=True)
        dist = torch.addmm(c2 + x2, xt, et.t(), alpha=-2.0, beta=1.0)
        dmin = dist.min(dim=1)[0].numpy()
        gap, idx64 = fp64_gap(xn, en, idx)
        out.update({tag + ".shape": np.array([N, D, K, seed], np.int64), tag + ".idx": idx, tag + ".dmin": dmin,
                    tag + ".x2": x2.view(-1).numpy(), tag + ".c2": c2.numpy(),
                    tag + ".gap64": gap, tag + ".idx64": idx64,
                    tag + ".xsum": np.array([np.float64(xn.astype(np.float64).sum()), np.float64(en.astype(np.float64).sum())])})
        print(tag, (N, D, K), "fp32-vs-fp64 index disagreements:", int((idx != idx64).sum()))
    # vq_st backward with a live codebook (index_add_ branch, vector_quantization.py:53-61)
    x = torch.randn(300, 16, generator=g).requires_grad_(True)
    e = torch.randn(24, 16, generator=g).requires_grad_(True)
    codes, idxf = ref_vq_st(x, e)
    w = torch.randn(300, 16, generator=g)
    (codes * w).sum().backward()
    out.update({"st.x": x.detach().numpy(), "st.e": e.detach().numpy(), "st.w": w.numpy(), "st.idx": idxf.numpy(),
                "st.codes": codes.detach().numpy(), "st.gx": x.grad.numpy(), "st.ge": e.grad.numpy()})
    np.savez_compressed(os.path.join(HERE, "vq_ops.npz"), **out)


def make_resblock_fixture():
    torch.manual_seed(3)
    blk = ref_models.ResBlock(8)
    blk.apply(ref_models.weights_init)
    with torch.no_grad():  # non-trivial affine + running stats so eval mode is exercised
        for m in blk.block:
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5)
                m.bias.uniform_(-0.3, 0.3)
                m.running_mean.uniform_(-0.2, 0.2)
                m.running_var.uniform_(0.5, 1.5)
            if isinstance(m, torch.nn.Conv2d):
                m.bias.uniform_(-0.1, 0.1)
    out = sd_np(blk)
    x0 = torch.randn(2, 8, 5, 6)
    blk.train()
    xt = x0.clone()
    y = blk(xt)
    out.update({"x": x0.numpy(), "y_train": y.detach().numpy(), "x_after_train": xt.numpy()})
    out.update({"sd_after." + k: v.detach().numpy().copy() for k, v in blk.state_dict().items()})
    blk.eval()
    xe = x0.clone()
    ye = blk(xe)
    out.update({"y_eval": ye.detach().numpy(), "x_after_eval": xe.numpy()})
    np.savez_compressed(os.path.join(HERE, "resblock.npz"), **out)


def make_model_fixture(name, dim, z_dim, shapes, full_grads, steps=1, dp=False):
    torch.manual_seed(1)  # main.py:43,71
    model = ref_models.VQVAE(1, dim, z_dim)
    out = sd_np(model, "sd0.")
    out["cfg"] = np.array([dim, z_dim], np.int64)
    g = torch.Generator().manual_seed(1234)
    for si, shp in enumerate(shapes):
        tag = "s%d." % si
        c = torch.rand(*shp, generator=g)  # normalised mel range [0,1)  (audio_tacotron.py:228-234)
        m = ref_models.VQVAE(1, dim, z_dim)
        m.load_state_dict(model.state_dict())
        opt = torch.optim.Adam(m.parameters(), lr=1e-3)  # main.py:124
        rec = ref_step(m, opt, c)
        out[tag + "c"] = c.numpy()
        out[tag + "losses"] = np.array([rec["loss_recons"].item(), rec["loss_vq"].item(), rec["loss_commit"].item()], np.float64)
        with torch.no_grad():
            z_ = rec["z_e"].permute(0, 2, 3, 1).contiguous()
            out[tag + "idx"] = ref_vq(z_, model.codebook.embedding.weight).numpy()
        if full_grads:
            out[tag + "x_tilde"] = rec["x_tilde"].numpy()
            out[tag + "z_e"] = rec["z_e"].numpy()
            out[tag + "z_q"] = rec["z_q"].numpy()
            for k, v in rec["grads"].items():
                out[tag + "grad." + k] = v.numpy()
            out.update(sd_np(m, tag + "sd1."))
        else:
            for k, v in rec["grads"].items():
                out[tag + "gnorm." + k] = np.array(v.double().norm().item())
            out[tag + "x_tilde_sum"] = np.array([rec["x_tilde"].double().sum().item(), rec["x_tilde"].double().abs().sum().item()])
        # eval-mode forward + encode/decode on the pre-step weights (test.py:73-106, models.py:188-196)
        if si == 0:
            model.eval()
            with torch.no_grad():
                xe, ze, zq = model(c)
                lat = model.encode(c)
                dec = model.decode(lat)
            out["eval.x_tilde" if full_grads else "eval.x_tilde_sum"] = xe.numpy() if full_grads else np.array([xe.double().sum().item()])
            out["eval.latents"] = lat.numpy()
            if full_grads:
                out["eval.decode"] = dec.numpy()
            out["eval.loss_vq"] = np.array(F.mse_loss(zq, ze).item())
            model.train()
        # multi-step trajectory on the fixed batch (F6)
        if si == 0 and steps > 1:
            traj = [out[tag + "losses"].copy()]
            for _ in range(steps - 1):
                r = ref_step(m, opt, c)
                traj.append(np.array([r["loss_recons"].item(), r["loss_vq"].item(), r["loss_commit"].item()]))
            out["traj.losses"] = np.stack(traj)
            out.update(sd_np(m, "traj.sd."))
    if dp:  # F7: two replicas on two shards, per-rank BN, averaged grads, one Adam step
        c0 = torch.rand(*shapes[0], generator=g)
        c1 = torch.rand(*shapes[0], generator=g)
        recs = []
        for c in (c0, c1):
            m = ref_models.VQVAE(1, dim, z_dim)
            m.load_state_dict(model.state_dict())
            recs.append((m, ref_step(m, torch.optim.Adam(m.parameters(), lr=1e-3), c)))
        m = ref_models.VQVAE(1, dim, z_dim)
        m.load_state_dict(model.state_dict())
        opt = torch.optim.Adam(m.parameters(), lr=1e-3)
        for k, p in m.named_parameters():
            p.grad = (recs[0][1]["grads"][k] + recs[1][1]["grads"][k]) / 2.0
            out["dp.grad." + k] = p.grad.numpy().copy()
        opt.step()
        out["dp.c0"], out["dp.c1"] = c0.numpy(), c1.numpy()
        out.update(sd_np(m, "dp.sd1."))
        out["dp.losses"] = np.array([[r["loss_recons"].item(), r["loss_vq"].item(), r["loss_commit"].item()] for _, r in recs])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, {k: float(v) for k, v in zip(("recons", "vq", "commit"), out["s0.losses"])})


if __name__ == "__main__":
    make_vq_fixtures()
    make_resblock_fixture()
    make_model_fixture("model_tiny", 16, 32, [(2, 1, 80, 64), (2, 1, 80, 31)], full_grads=True, steps=5, dp=True)
    make_model_fixture("model_cfg1", 64, 128, [(2, 1, 80, 64)], full_grads=False)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
