"""The step on the far side of training: evaluation-mode loss, reconstruction dump and the
checkpoint layout, restated from the reference's src/test.py:73-106 and src/main.py:61-66,137-163,216-220.

  * `test_vqvae(args, model, test_loader, device, epoch)` -- same signature and behaviour as the reference's function
    (ljspeech branch): model.eval(), forward, zero-pad the reconstruction to the input width, accumulate
    mse(target, c) and mse(z_q, z_e) over the batches, divide by the number of batches, print their sum.
    Returns (loss_recons, loss_vq) as floats (the reference returns nothing).
  * `eval_losses(model, c)` -- the same two numbers for one batch with no autograd and no host round trip: the
    HIP forward stacks and the fused loss kernels (nsg_mse_padded folds the zero-pad).
  * `export_reconstruction(model, c, path)` -- main.py:150-163: x_tilde.squeeze(1) as a float32 .npy of shape (B, 80, T').
  * `checkpoint_state` / `save_checkpoint` / `load_checkpoint` -- main.py:61-66,216-220: {'epoch', 'arch',
    'state_dict', 'optimizer'}; state_dict keys are the reference's, the optimiser state is in torch.optim.Adam's
    layout (FlatAdam.state_dict), so checkpoints move both ways between the reference and this package.
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.nn.functional as F

from . import engine, functional as Fn, ops


@torch.no_grad()
def eval_losses(model, c: torch.Tensor):
    """c (B,1,80,T) on the GPU -> (loss_recons, loss_vq) device scalars, eval-mode statistics (running mean / var)."""
    x = Fn.to_nhwc(c)
    B, H, T, _ = x.shape
    dtype = getattr(model, "compute_dtype", torch.float32)
    encP, decP = engine.encoder_params(model.encoder), engine.decoder_params(model.decoder)
    ze, _ = engine.encoder_forward(x, encP, False, dtype=dtype)
    D = ze.shape[-1]
    cb = model.codebook.embedding.weight.detach()
    _, zq, _ = ops.vq_forward(ze.view(-1, D), cb, want_codes=True, impl=getattr(model.codebook, "search_impl", "mfma"))
    zq = zq.view_as(ze)
    xt, _ = engine.decoder_forward(zq, decP, False, dtype=dtype)
    loss_recons, _ = ops.mse_padded(xt, x, B * H, xt.shape[2], T, want_grad=False)
    loss_vq, _, _ = ops.vq_losses(ze, zq, want_dz=False, want_dq=False)
    return loss_recons[0], loss_vq[0]


def test_vqvae(args, model, test_loader, device, epoch):
    """Drop-in for src/test.py:73-106 (ljspeech branch): `test_loader` yields (x, y, c, g, input_lengths), c (B, 80, T)."""
    model.eval()
    loss_recons = torch.zeros((), device=device)
    loss_vq = torch.zeros((), device=device)
    n = 0
    with torch.no_grad():
        for step, (x, y, c, g, input_lengths) in enumerate(test_loader):
            c = c.to(device).unsqueeze(1)
            x_tilde, z_e_x, z_q_x = model(c)
            target = F.pad(x_tilde, (0, c.size(3) - x_tilde.size(3))) if x_tilde.size(3) != c.size(3) else x_tilde
            loss_recons += F.mse_loss(target, c)
            loss_vq += F.mse_loss(z_q_x, z_e_x)
            n += 1
    if n == 0:
        raise ValueError("test_vqvae: empty loader")
    loss_recons, loss_vq = float(loss_recons / n), float(loss_vq / n)
    print('====> Test set loss: {:.4f}'.format(loss_recons + loss_vq))
    return loss_recons, loss_vq


__test__ = False  # (not a pytest module even though a function is named test_*)


@torch.no_grad()
def export_reconstruction(model, c: torch.Tensor, path: str) -> np.ndarray:
    """main.py:150-163: reconstruction of one batch as a float32 array (B, 80, T') saved with np.save."""
    was_training = model.training
    model.eval()
    x_tilde, _, _ = model(c)
    model.train(was_training)
    rec = x_tilde.squeeze(1).float().cpu().numpy()
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    np.save(path, rec, allow_pickle=False)
    return rec


def checkpoint_state(epoch: int, arch: str, model, optimizer) -> dict:
    return {"epoch": epoch, "arch": arch, "state_dict": model.state_dict(), "optimizer": optimizer.state_dict()}


def checkpoint_filename(args) -> str:
    """main.py:61-65 (the reference never creates the directory; save_checkpoint here does)."""
    return './models/{}/checkpoint_{}_{}_{}.pth.tar'.format(args.model, args.dataset, args.dim, args.z_dim)


def save_checkpoint(args, state, filename: str | None = None) -> str:
    filename = filename or checkpoint_filename(args)
    os.makedirs(os.path.dirname(os.path.abspath(filename)), exist_ok=True)
    torch.save(state, filename)
    return filename


def load_checkpoint(filename: str, model, optimizer=None, map_location=None) -> dict:
    """Loads a checkpoint written by this package or by the reference's main.py (tensors only: weights_only=True)."""
    state = torch.load(filename, map_location=map_location, weights_only=True)
    model.load_state_dict(state["state_dict"])
    if optimizer is not None and state.get("optimizer") is not None:
        optimizer.load_state_dict(state["optimizer"])
    return state
