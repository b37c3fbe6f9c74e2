"""The latent prior of the model family: the reference's GatedPixelCNN over the grid of code indices
(src/models.py:219-341), on the HIP kernels of the main path.

Same class names, constructor arguments, attributes and state_dict keys as the reference (GatedActivation,
GatedMaskedConv2d, GatedPixelCNN), same initialisation (identical RNG consumption: `torch.manual_seed(s);
GatedPixelCNN(...)` reproduces the reference's initial weights bit for bit, tests/test_host_logic.py).

How it maps onto the kernels:
  * every convolution is nsg_conv_forward / nsg_conv_dgrad / nsg_conv_wgrad.  The masked stacks use rectangular kernels
    with pad-then-crop -- vertical (k//2+1, k) padded (k//2, k//2) and cropped to H rows, horizontal (1, k//2+1) padded
    (0, k//2) and cropped to W columns (models.py:238-252,268-273): the conv descriptor's rectangular form (k, k_w, pad,
    pad_w, cropped OH / OW), so only the taps that exist are computed;
  * GatedActivation with the class-conditional add is one kernel each way (nsg_gated_activation_*);
  * the two embeddings are nsg_gather_rows / nsg_index_add_rows; the cross-entropy of the logits is nsg_cross_entropy.
All tensors between kernels are fp32 NHWC rows; module inputs / outputs keep the reference's NCHW logical shapes
(channels_last views).

One deliberate generalisation (also in oracle/pixelcnn_oracle.py): the reference crops the vertical stack's rows with the
input WIDTH and the horizontal stack's columns with the input HEIGHT (models.py:269,273), so it only runs on square grids;
here rows are cropped to the height and columns to the width -- identical on square grids, and what the (20, T/4) grid of
the VQ-VAE's codes needs.  `generate` is the evident intent of the reference's (which passes a nested tuple to
torch.zeros, models.py:325-341).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Function

from . import functional as Fn, ops
from .models import weights_init
from .vector_quantization import codebook_lookup


class _Conv(Function):
    """y = conv2d(x, w, b) on NHWC rows, stride 1, kernel (kh, kw) = w's own extent, padding (ph, pw), the output cropped to
    the input's (H, W) at the bottom / right -- the reference's pad-then-crop of the masked stacks (models.py:268-273) and,
    for odd square kernels with pad = k // 2, plain 'same' convolution.  Optional ReLU of the output."""

    @staticmethod
    def forward(ctx, x, w, b, pad, relu_out):
        x = x.contiguous()
        B, H, W, Ci = x.shape
        Co, _, kh, kw = w.shape
        d = ops.conv_desc(B, H, W, Ci, Co, (kh, kw), 1, pad, out_hw=(H, W))
        wf, wd = ops.pack_weights(d, w.detach().contiguous())
        y = ops.conv_forward(d, x, wf, b.detach(), flags=ops.NSG_RELU_OUT if relu_out else 0)
        ctx.d, ctx.wd, ctx.relu_out, ctx.wshape = d, wd, relu_out, tuple(w.shape)
        ctx.save_for_backward(x, y if relu_out else None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, y = ctx.saved_tensors
        gy = gy.contiguous()
        if ctx.relu_out:
            gy = ops.relu_backward_add(gy, None, y)
        dw, db = ops.conv_wgrad(ctx.d, x, gy, ctx.wshape)
        dx = ops.conv_dgrad(ctx.d, gy, ctx.wd) if ctx.needs_input_grad[0] else None
        return dx, dw, db, None, None


class _Gate(Function):
    """tanh(a) * sigmoid(b) of the channel halves of x (+ per-clip conditioning rows)."""

    @staticmethod
    def forward(ctx, x, cond):
        x = x.contiguous()
        cond = cond.contiguous() if cond is not None else None
        ctx.save_for_backward(x, cond)
        return ops.gated_activation(x, cond)

    @staticmethod
    def backward(ctx, gy):
        x, cond = ctx.saved_tensors
        dx = ops.gated_activation_backward(x, cond, gy.contiguous())
        dcond = ops.clip_colsum(dx, cond.shape[0]) if (cond is not None and ctx.needs_input_grad[1]) else None
        return dx, dcond


class _Add(Function):
    @staticmethod
    def forward(ctx, a, b):
        return ops.add(a.contiguous(), b.contiguous())

    @staticmethod
    def backward(ctx, g):
        return g, g


class _CrossEntropy(Function):
    @staticmethod
    def forward(ctx, logits2d, target):
        loss, dl = ops.cross_entropy(logits2d.contiguous(), target.contiguous(), want_grad=True)
        ctx.save_for_backward(dl)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return dl * g, None


def _conv(x_nhwc, conv: nn.Conv2d, relu_out: bool = False):
    """The module's own (possibly rectangular) kernel and padding, output cropped to the input extent."""
    return _Conv.apply(x_nhwc, conv.weight, conv.bias, tuple(conv.padding), relu_out)


class GatedActivation(nn.Module):
    def forward(self, x):
        return Fn.to_nchw_view(_Gate.apply(Fn.to_nhwc(x), None))


class GatedMaskedConv2d(nn.Module):
    def __init__(self, mask_type, dim, kernel, residual=True, n_classes=10):
        super().__init__()
        assert kernel % 2 == 1, "Kernel size must be odd"
        self.mask_type = mask_type
        self.residual = residual
        self.kernel = kernel
        self.class_cond_embedding = nn.Embedding(n_classes, 2 * dim)
        self.vert_stack = nn.Conv2d(dim, dim * 2, (kernel // 2 + 1, kernel), 1, (kernel // 2, kernel // 2))
        self.vert_to_horiz = nn.Conv2d(2 * dim, 2 * dim, 1)
        self.horiz_stack = nn.Conv2d(dim, dim * 2, (1, kernel // 2 + 1), 1, (0, kernel // 2))
        self.horiz_resid = nn.Conv2d(dim, dim, 1)
        self.gate = GatedActivation()

    def make_causal(self):
        self.vert_stack.weight.data[:, :, -1].zero_()       # mask the final row
        self.horiz_stack.weight.data[:, :, :, -1].zero_()   # mask the final column

    def forward_nhwc(self, x_v, x_h, h):
        """x_v, x_h (B, H, W, dim) NHWC rows, h (B,) int64 class labels -> (out_v, out_h) NHWC."""
        if self.mask_type == 'A':
            self.make_causal()
        cond = codebook_lookup(self.class_cond_embedding.weight, h.view(-1))          # (B, 2 dim)
        h_vert = _conv(x_v, self.vert_stack)
        out_v = _Gate.apply(h_vert, cond)
        h_horiz = _conv(x_h, self.horiz_stack)
        v2h = _conv(h_vert, self.vert_to_horiz)
        out = _Gate.apply(_Add.apply(v2h, h_horiz), cond)
        out_h = _conv(out, self.horiz_resid)
        if self.residual:
            out_h = _Add.apply(out_h, x_h)
        return out_v, out_h

    def forward(self, x_v, x_h, h):
        out_v, out_h = self.forward_nhwc(Fn.to_nhwc(x_v), Fn.to_nhwc(x_h), h)
        return Fn.to_nchw_view(out_v), Fn.to_nchw_view(out_h)


class GatedPixelCNN(nn.Module):
    def __init__(self, input_dim=256, dim=64, n_layers=15, n_classes=10):
        super().__init__()
        if dim % 4 != 0:
            raise ValueError("GatedPixelCNN: dim must be a multiple of 4 on this path")
        self.dim = dim
        self.embedding = nn.Embedding(input_dim, dim)
        self.layers = nn.ModuleList()
        for i in range(n_layers):
            mask_type = 'A' if i == 0 else 'B'
            kernel = 7 if i == 0 else 3
            residual = False if i == 0 else True
            self.layers.append(GatedMaskedConv2d(mask_type, dim, kernel, residual, n_classes))
        self.output_conv = nn.Sequential(nn.Conv2d(dim, 512, 1), nn.ReLU(True), nn.Conv2d(512, input_dim, 1))
        self.apply(_weights_init_quiet)

    def forward_nhwc(self, x, label):
        """x int64 (B, H, W), label int64 (B,) -> logits (B, H, W, input_dim) NHWC rows."""
        B, H, W = x.shape
        e = codebook_lookup(self.embedding.weight, x.reshape(-1)).view(B, H, W, self.dim)
        x_v, x_h = e, e
        for layer in self.layers:
            x_v, x_h = layer.forward_nhwc(x_v, x_h, label)
        y = _conv(x_h, self.output_conv[0], relu_out=True)     # the ReLU is fused into the conv's store
        return _conv(y, self.output_conv[2])

    def forward(self, x, label):
        return Fn.to_nchw_view(self.forward_nhwc(x, label))

    def loss(self, x, label):
        """Mean cross-entropy of the prior's logits against the codes themselves (each code from its causal context)."""
        logits = self.forward_nhwc(x, label)
        return _CrossEntropy.apply(logits.view(-1, logits.shape[-1]), x.reshape(-1))

    @torch.no_grad()
    def generate(self, label, shape=(8, 8), batch_size=64):
        """Ancestral sampling in raster order (the intent of models.py:325-341)."""
        param = next(self.parameters())
        x = torch.zeros((batch_size,) + tuple(shape), dtype=torch.int64, device=param.device)
        for i in range(shape[0]):
            for j in range(shape[1]):
                logits = self.forward_nhwc(x, label)
                probs = F.softmax(logits[:, i, j, :], -1)
                x[:, i, j] = probs.multinomial(1).squeeze(-1)
        return x


def _weights_init_quiet(m):
    """models.weights_init (src/models.py:25-32) without the reference's "Skipping initialization of ..." print for the
    gated layers (their class name contains 'Conv' but they own no .weight: the reference skips them too)."""
    if isinstance(m, GatedMaskedConv2d):
        return
    weights_init(m)
