"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the reference's latent prior, GatedPixelCNN
(src/models.py:219-341), as plain functions over a state dict.  Pinned to tests/golden/prior_tiny.npz, which
tests/golden/make_golden_prior.py generates by importing the reference's own class.

One deliberate generalisation: the reference crops the vertical stack's rows with the input WIDTH and the horizontal
stack's columns with the input HEIGHT (models.py:269,273), so it only runs on square grids; here rows are cropped to the
height and columns to the width, which is the same thing on a square grid (the pinned case) and what the masks require
on the (20, T/4) grid of the VQ-VAE's codes.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def gate(x):
    """GatedActivation, models.py:219-226."""
    a, b = x.chunk(2, dim=1)
    return torch.tanh(a) * torch.sigmoid(b)


def state_keys(input_dim: int, dim: int, n_layers: int, n_classes: int):
    """[(name, shape)] of GatedPixelCNN.state_dict() in order (models.py:228-312)."""
    keys = [("embedding.weight", (input_dim, dim))]
    for i in range(n_layers):
        k = 7 if i == 0 else 3
        p = f"layers.{i}."
        keys += [(p + "class_cond_embedding.weight", (n_classes, 2 * dim)),
                 (p + "vert_stack.weight", (2 * dim, dim, k // 2 + 1, k)), (p + "vert_stack.bias", (2 * dim,)),
                 (p + "vert_to_horiz.weight", (2 * dim, 2 * dim, 1, 1)), (p + "vert_to_horiz.bias", (2 * dim,)),
                 (p + "horiz_stack.weight", (2 * dim, dim, 1, k // 2 + 1)), (p + "horiz_stack.bias", (2 * dim,)),
                 (p + "horiz_resid.weight", (dim, dim, 1, 1)), (p + "horiz_resid.bias", (dim,))]
    keys += [("output_conv.0.weight", (512, dim, 1, 1)), ("output_conv.0.bias", (512,)),
             ("output_conv.2.weight", (input_dim, 512, 1, 1)), ("output_conv.2.bias", (input_dim,))]
    return keys


def layer(st, i, x_v, x_h, label, n_layers_unused=None):
    """GatedMaskedConv2d.forward, models.py:263-283.  Layer 0 is mask 'A', kernel 7, no residual; the rest 'B', 3, residual.
    NOTE: mask 'A' zeroes st's weights in place, exactly as make_causal does (models.py:259-261)."""
    p = f"layers.{i}."
    k = 7 if i == 0 else 3
    if i == 0:
        st[p + "vert_stack.weight"].data[:, :, -1].zero_()
        st[p + "horiz_stack.weight"].data[:, :, :, -1].zero_()
    H, W = x_v.shape[-2], x_v.shape[-1]
    h = F.embedding(label, st[p + "class_cond_embedding.weight"])
    h_vert = F.conv2d(x_v, st[p + "vert_stack.weight"], st[p + "vert_stack.bias"], 1, (k // 2, k // 2))[:, :, :H, :]
    out_v = gate(h_vert + h[:, :, None, None])
    h_horiz = F.conv2d(x_h, st[p + "horiz_stack.weight"], st[p + "horiz_stack.bias"], 1, (0, k // 2))[:, :, :, :W]
    v2h = F.conv2d(h_vert, st[p + "vert_to_horiz.weight"], st[p + "vert_to_horiz.bias"])
    out = gate(v2h + h_horiz + h[:, :, None, None])
    out_h = F.conv2d(out, st[p + "horiz_resid.weight"], st[p + "horiz_resid.bias"])
    if i > 0:
        out_h = out_h + x_h
    return out_v, out_h


def forward(st, x, label, n_layers):
    """GatedPixelCNN.forward, models.py:314-323: x int64 (B, H, W), label int64 (B,) -> logits (B, input_dim, H, W)."""
    e = F.embedding(x, st["embedding.weight"]).permute(0, 3, 1, 2)
    x_v, x_h = e, e
    for i in range(n_layers):
        x_v, x_h = layer(st, i, x_v, x_h, label)
    y = F.conv2d(x_h, st["output_conv.0.weight"], st["output_conv.0.bias"])
    y = F.relu(y)
    return F.conv2d(y, st["output_conv.2.weight"], st["output_conv.2.bias"])


def loss_and_grads(st, x, label, n_layers):
    """Cross-entropy of the prior's logits against the codes themselves, and d loss / d every parameter."""
    params = {k: v.clone().requires_grad_(True) for k, v in st.items()}
    logits = forward(params, x, label, n_layers)
    loss = F.cross_entropy(logits, x)
    grads = torch.autograd.grad(loss, list(params.values()))
    return logits.detach(), loss.detach(), dict(zip(params.keys(), grads)), {k: v.detach() for k, v in params.items()}
