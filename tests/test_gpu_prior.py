"""GPU parity tests of the latent prior (GatedPixelCNN) against the fixture generated from the
reference's own class (tests/golden/prior_tiny.npz) and against the CPU oracle (oracle/pixelcnn_oracle.py)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from neural_sound_generation_amd import ops  # noqa: E402
from neural_sound_generation_amd.prior import GatedPixelCNN, GatedActivation  # noqa: E402
from oracle import pixelcnn_oracle as P  # noqa: E402

DEV = "cuda:0"


def build(g):
    input_dim, dim, n_layers, n_classes = (int(v) for v in g["cfg"])
    m = GatedPixelCNN(input_dim, dim, n_layers, n_classes)
    m.load_state_dict({k[4:]: torch.from_numpy(np.array(g[k])) for k in g.files if k.startswith("sd0.")})
    return m.to(DEV), n_layers


def test_gated_activation_and_cross_entropy_kernels():
    gen = torch.Generator().manual_seed(0)
    x = torch.randn(3, 5, 7, 24, generator=gen)
    cond = torch.randn(3, 24, generator=gen)
    xr, cr = x.clone().requires_grad_(True), cond.clone().requires_grad_(True)
    a, b = (xr + cr[:, None, None, :]).chunk(2, dim=-1)
    y = torch.tanh(a) * torch.sigmoid(b)
    dy = torch.randn(y.shape, generator=gen)
    gx, gc = torch.autograd.grad(y, [xr, cr], dy)
    yg = ops.gated_activation(x.to(DEV), cond.to(DEV))
    np.testing.assert_allclose(yg.cpu().numpy(), y.detach().numpy(), rtol=1e-5, atol=1e-6)
    dxg = ops.gated_activation_backward(x.to(DEV), cond.to(DEV), dy.to(DEV))
    np.testing.assert_allclose(dxg.cpu().numpy(), gx.numpy(), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(ops.clip_colsum(dxg, 3).cpu().numpy(), gc.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(ops.gated_activation(x.to(DEV)).cpu().numpy(),
                               (torch.tanh(x[..., :12]) * torch.sigmoid(x[..., 12:])).numpy(), rtol=1e-5, atol=1e-6)
    got = GatedActivation()(x.permute(0, 3, 1, 2).to(DEV))                  # the module takes NCHW like the reference's
    np.testing.assert_allclose(got.cpu().numpy(), P.gate(x.permute(0, 3, 1, 2)).numpy(), rtol=1e-5, atol=1e-6)
    for M, K in ((37, 512), (64, 10), (5, 100)):
        l = (torch.randn(M, K, generator=gen) * 3).requires_grad_(True)
        t = torch.randint(0, K, (M,), generator=gen)
        want = F.cross_entropy(l, t)
        (gl,) = torch.autograd.grad(want, [l])
        loss, dl = ops.cross_entropy(l.detach().to(DEV), t.to(DEV))
        assert abs(loss.item() - want.item()) <= 1e-6 * abs(want.item()) + 1e-7
        np.testing.assert_allclose(dl.cpu().numpy(), gl.numpy(), rtol=1e-4, atol=1e-7)


def test_prior_forward_loss_and_gradients_match_the_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "prior_tiny.npz"))
    model, n_layers = build(g)
    x, label = torch.from_numpy(g["x"]).to(DEV), torch.from_numpy(g["label"]).to(DEV)
    logits = model(x, label)
    assert tuple(logits.shape) == g["logits"].shape                                     # (B, input_dim, H, W) like the reference
    np.testing.assert_allclose(logits.detach().cpu().numpy(), g["logits"], rtol=1e-4, atol=2e-5)
    for k, v in model.state_dict().items():                                              # make_causal zeroed layer 0's last row / column
        assert np.array_equal(v.cpu().numpy(), g["sd1." + k]), k
    loss = model.loss(x, label)
    assert abs(loss.item() - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    loss.backward()
    for k, p in model.named_parameters():
        want = g["grad." + k]
        np.testing.assert_allclose(p.grad.cpu().numpy(), want, rtol=2e-3, atol=2e-5 * max(1.0, float(np.abs(want).max())), err_msg=k)
    # the user-composed loss (F.cross_entropy on the module's NCHW logits) differentiates through the same stacks
    model.zero_grad()
    F.cross_entropy(model(x, label), x).backward()
    np.testing.assert_allclose(model.embedding.weight.grad.cpu().numpy(), g["grad.embedding.weight"], rtol=2e-3, atol=2e-5)
    # causality witness from the reference run
    l2 = model(torch.from_numpy(g["x2"]).to(DEV), label)
    np.testing.assert_allclose(l2.detach().cpu().numpy(), g["logits2"], rtol=1e-4, atol=2e-5)
    same = (l2.detach() - logits.detach()).abs().amax(dim=(0, 1)).cpu() == 0
    assert bool(same[:3].all()) and bool(same[3, :5].all())


def test_prior_on_the_latent_grid_and_sampling(golden_dir):
    """The (20, T/4) grid is not square: the reference cannot run it (models.py:269,273); the oracle's generalisation is the check."""
    g = np.load(os.path.join(golden_dir, "prior_tiny.npz"))
    model, n_layers = build(g)
    input_dim = int(g["cfg"][0])
    st = {k[4:]: torch.from_numpy(np.array(g[k])) for k in g.files if k.startswith("sd0.")}
    gen = torch.Generator().manual_seed(3)
    x = torch.randint(0, input_dim, (2, 20, 16), generator=gen)
    label = torch.randint(0, int(g["cfg"][3]), (2,), generator=gen)
    want_logits, want_loss, want_grads, _ = P.loss_and_grads(st, x, label, n_layers)
    logits = model(x.to(DEV), label.to(DEV))
    np.testing.assert_allclose(logits.detach().cpu().numpy(), want_logits.numpy(), rtol=1e-4, atol=2e-5)
    loss = model.loss(x.to(DEV), label.to(DEV))
    assert abs(loss.item() - want_loss.item()) <= 1e-5 * abs(want_loss.item())
    loss.backward()
    for k, p in model.named_parameters():
        w = want_grads[k].numpy()
        np.testing.assert_allclose(p.grad.cpu().numpy(), w, rtol=2e-3, atol=2e-5 * max(1.0, float(np.abs(w).max())), err_msg=k)
    torch.manual_seed(0)
    s = model.generate(label.to(DEV), shape=(3, 5), batch_size=2)
    assert tuple(s.shape) == (2, 3, 5) and s.dtype == torch.int64 and int(s.min()) >= 0 and int(s.max()) < input_dim
    # a few optimiser steps on the prior's own loss reduce it
    opt = torch.optim.Adam(model.parameters(), lr=3e-3)
    first = None
    for _ in range(10):
        opt.zero_grad()
        l = model.loss(x.to(DEV), label.to(DEV))
        l.backward()
        opt.step()
        first = first if first is not None else l.item()
    assert l.item() < first
