"""Generates tests/golden/prior_tiny.npz by IMPORTING the reference's GatedPixelCNN (src/models.py:219-341).

Runs only in the build container (needs /root/reference); the GPU box uses the committed .npz.
    python tests/golden/make_golden_prior.py

The reference never trains this model (it is not wired into main.py), so the step recorded here is the obvious one:
logits = model(x, label); loss = F.cross_entropy(logits, x) (the prior predicts each code from its causal context);
loss.backward().  Note that the first layer's forward zeroes part of its own weights in place (make_causal, :259-261):
the state dict is saved before AND after the forward.
"""
import io
import os
import sys
import contextlib

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference/src")
import models as ref_models  # noqa: E402

torch.set_num_threads(8)


def main():
    cfg = dict(input_dim=32, dim=16, n_layers=3, n_classes=4)
    torch.manual_seed(1)
    with contextlib.redirect_stdout(io.StringIO()):          # weights_init prints "Skipping initialization of ..." per gated layer
        model = ref_models.GatedPixelCNN(**cfg)
    out = {"cfg": np.array([cfg["input_dim"], cfg["dim"], cfg["n_layers"], cfg["n_classes"]])}
    for k, v in model.state_dict().items():
        out["sd0." + k] = v.detach().numpy().copy()
    g = torch.Generator().manual_seed(7)
    x = torch.randint(0, cfg["input_dim"], (2, 8, 8), generator=g)   # square: the reference crops rows by W and columns by H (models.py:269,273)
    label = torch.randint(0, cfg["n_classes"], (2,), generator=g)
    logits = model(x, label)
    loss = F.cross_entropy(logits, x)
    loss.backward()
    out["x"], out["label"] = x.numpy(), label.numpy()
    out["logits"] = logits.detach().numpy()
    out["loss"] = np.array(loss.item(), dtype=np.float64)
    for k, p in model.named_parameters():
        out["grad." + k] = p.grad.detach().numpy().copy()
    for k, v in model.state_dict().items():
        out["sd1." + k] = v.detach().numpy().copy()
    # causality witness: changing pixel (i, j) must not change logits at or before (i, j) in raster order
    x2 = x.clone()
    x2[:, 3, 4] = (x2[:, 3, 4] + 1) % cfg["input_dim"]
    with torch.no_grad():
        l2 = model(x2, label)
    out["x2"], out["logits2"] = x2.numpy(), l2.numpy()
    path = os.path.join(HERE, "prior_tiny.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes; loss", loss.item())


if __name__ == "__main__":
    main()
