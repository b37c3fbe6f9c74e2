"""Explicit forward / backward of the VQ-VAE encoder, decoder and ResBlock over NHWC buffers.

This is the static "graph" of the hot path (reference: src/models.py:145-216), written as straight
sequences of C-ABI kernel calls with the fusion decisions made by hand:
  * the ReLU that heads a ResBlock / precedes a conv is fused into that conv's operand load;
  * BatchNorm normalise + ReLU (+ the ResBlock skip add of the ReLU'd input) is one kernel;
  * BatchNorm backward folds the ReLU mask; the ResBlock's input gradient folds add + ReLU mask.
No autograd runs inside; neural_sound_generation_amd.functional wraps these pieces in
torch.autograd.Function so they compose with user code, and train.py drives them directly.

Parameter bundles are plain tuples of tensors in the reference's state_dict order.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch

from . import ops
from .ops import NSG_RELU_IN, NSG_RELU_OUT, NSG_TANH_OUT


@dataclass
class BNParams:
    weight: torch.Tensor
    bias: torch.Tensor
    running_mean: torch.Tensor
    running_var: torch.Tensor
    num_batches_tracked: Optional[torch.Tensor] = None


@dataclass
class ConvParams:
    weight: torch.Tensor
    bias: torch.Tensor


@dataclass
class ResBlockParams:
    conv1: ConvParams   # block.1  3x3
    bn1: BNParams       # block.2
    conv2: ConvParams   # block.4  1x1
    bn2: BNParams       # block.5


@dataclass
class EncoderParams:
    conv0: ConvParams   # encoder.0  Conv2d(1, D, 4, 2, 1)
    bn0: BNParams       # encoder.1
    conv3: ConvParams   # encoder.3  Conv2d(D, D, 4, 2, 1)
    res4: ResBlockParams
    res5: ResBlockParams


@dataclass
class DecoderParams:
    res0: ResBlockParams
    res1: ResBlockParams
    convt3: ConvParams  # decoder.3  ConvTranspose2d(D, D, 4, 2, 1)
    bn4: BNParams       # decoder.4
    convt6: ConvParams  # decoder.6  ConvTranspose2d(D, 1, 4, 2, 1)


# BatchNorm batch statistics from the conv epilogue (nsg_conv_forward_bnstats) or from a separate
# pass over the conv output (nsg_bn_stats).  Measured on MI355X in fp32 (B=64, D=128): separate pass
# 30.25 ms/step, fused epilogue 30.53 ms/step -- the GEMM epilogue sits on the MFMA-bound critical
# path while the stand-alone pass streams at 5 TB/s -- so the separate pass is the default.
import os as _os
FUSED_BN_STATS = _os.environ.get("NSG_FUSED_BN_STATS", "0") == "1"
# encoder.0-2 (Conv2d(1, D, 4, 2, 1) -> BatchNorm -> ReLU) as one operator that never stores the conv output
# (ops.c1conv_bn_relu_*): measured on MI355X, bf16, B=128: see DESIGN.md section 3.  NSG_FUSED_C1_LAYER=0 restores
# the four separate operators (same values: the fused passes recompute the conv output bit for bit).
FUSED_C1_LAYER = _os.environ.get("NSG_FUSED_C1_LAYER", "1") == "1"
# decoder.4-7 (BatchNorm -> ReLU -> ConvTranspose2d(D, 1, 4, 2, 1) -> Tanh) as one operator that stores neither the activated
# tensor nor the transposed conv's data gradient (ops.bn_relu_c1convt_*; bf16 tensors, D = 32..128).  NSG_FUSED_OUT_LAYER=0
# restores the separate operators.
FUSED_OUT_LAYER = _os.environ.get("NSG_FUSED_OUT_LAYER", "1") == "1"
# The ResBlock's 1x1 conv as a flat GEMM with the BatchNorm arithmetic of its neighbours in the operand staging
# (ops.bn_relu_conv1x1_* / ops.bn_backward_conv1x1_dgrad; bf16 tensors, dim = 32, 64, 128): relu(bn1(h1)) is never stored
# and bn2's input gradient is produced and consumed in one pass.  NSG_FUSED_1X1=0 restores the separate operators.
FUSED_1X1 = _os.environ.get("NSG_FUSED_1X1", "1") == "1"


# num_batches_tracked += 1 per BatchNorm is ten tiny launches a step; a fused step collects the counters here
# and bumps them with one launch of the library's own kernel (deferred_batch_counters -> nsg_increment_counters)
_nbt_pending = None


class deferred_batch_counters:
    def __enter__(self):
        global _nbt_pending
        self._outer = _nbt_pending
        _nbt_pending = []
        return self

    def __exit__(self, *exc):
        global _nbt_pending
        pending, _nbt_pending = _nbt_pending, self._outer
        if pending and exc[0] is None:
            ops.increment_counters(pending)
        return False


def _bump(bn: "BNParams"):
    if bn.num_batches_tracked is None:
        return
    if _nbt_pending is not None:
        _nbt_pending.append(bn.num_batches_tracked)
    else:
        ops.increment_counters([bn.num_batches_tracked])


FUSED_OUT_LOSS = _os.environ.get("NSG_FUSED_OUT_LOSS", "1") == "1"  # reconstruction loss + Tanh backward in the output layer's image pass
FUSED_1X1_BWD = _os.environ.get("NSG_FUSED_1X1_BWD", "1") == "1"    # the 1x1 conv's data and weight gradients in one kernel (C = 128)
# BatchNorm batch statistics of a patch_gemm layer's output from that kernel's (wave-private) store phase instead of a read pass
# over the tensor: 1 = every such layer in front of a BatchNorm (the ResBlocks' 3x3 convs, decoder.3), 2 = decoder.3 only (the
# high-resolution tensor), 0 = off.  Numbers: DESIGN.md section 6.
# Measured in the step (same box, alternating): off 8.42 / 8.45 ms, decoder.3 only 8.26 / 8.35, all five 8.33 / 8.26 -- while the
# conv kernel's in-region rate is 978 / 974, 974 / 960 and 927 / 935 TF: the one high-resolution pass is nearly all of the gain.
# End of round 3 (scripts/ab_env.sh, six alternating pairs): all five 7.97-8.00 ms against 8.02-8.07 for decoder.3 only (0.6 %,
# inside the box-to-box spread) at a 4 % lower rate of the dominant kernel; the default stays at 2.
PATCH_BN_STATS = int(_os.environ.get("NSG_PATCH_BN_STATS", "2"))


def _patch_stats_ok(d, flags) -> bool:
    return (PATCH_BN_STATS and (PATCH_BN_STATS == 1 or d.transposed) and d.dtype == ops.NSG_BF16 and flags == 0 and d.C_in % 64 == 0
            and d.C_out % 128 == 0 and d.k_w == 0 and ((d.k == 3 and d.stride == 1) or (d.k == 4 and d.stride == 2)))


def _conv_bn(d, x, wf, conv: ConvParams, bn: BNParams, training: bool, flags=0):
    """conv (+fused input ReLU) followed by BatchNorm statistics: one fused call in training mode
    (the statistics come out of the conv epilogue), two calls in eval mode."""
    if training and (FUSED_BN_STATS or _patch_stats_ok(d, flags)):
        h, mean, invstd = ops.conv_forward_bnstats(d, x, wf, conv.bias, flags=flags, running_mean=bn.running_mean,
                                                   running_var=bn.running_var)
        _bump(bn)
        return h, mean, invstd
    h = ops.conv_forward(d, x, wf, conv.bias, flags=flags)
    mean, invstd = _bn_forward(h, bn, training)
    return h, mean, invstd


def _bn_forward(h, bn: BNParams, training: bool):
    C = bn.weight.numel()
    if training:
        mean, invstd = ops.bn_stats(h, C, bn.running_mean, bn.running_var)
        _bump(bn)
    else:
        mean, invstd = ops.bn_eval_stats(bn.running_mean, bn.running_var)
    return mean, invstd


# ------------------------------------------------------------------------------------------------
# ResBlock   y = relu(x) + BN(conv1x1(relu(BN(conv3x3(relu(x))))))      (src/models.py:145-158)
# ------------------------------------------------------------------------------------------------
def resblock_forward(r, P: ResBlockParams, training: bool, out_dtype=None, relu_out=False, packs=None):
    """r = relu(x) NHWC of the compute dtype, ALREADY ReLU'd by its producer: the reference's block starts
    with an in-place ReLU that overwrites its input (models.py:149), so nothing ever needs the un-ReLU'd
    tensor and the backward mask (x > 0) equals (r > 0).  Applying that ReLU where the tensor is written
    removes all ReLU work from the GEMM operand staging.
    out_dtype: storage type of the output (default: r's) -- the encoder's last block writes fp32 for the
    quantiser.  relu_out: store relu(y) because the consumer is again a ResBlock / the decoder's ReLU.
    packs: ((w_fwd, w_dgrad) of conv1, of conv2) when the caller packed the weights already (pack_all)."""
    B, H, W, D = r.shape
    d1 = ops.conv_desc(B, H, W, D, D, 3, 1, 1, dtype=r.dtype)
    d2 = ops.conv_desc(B, H, W, D, D, 1, 1, 0, dtype=r.dtype)
    (wf1, wd1), (wf2, wd2) = packs if packs is not None else (ops.pack_weights(d1, P.conv1.weight), ops.pack_weights(d2, P.conv2.weight))
    h1, m1, i1 = _conv_bn(d1, r, wf1, P.conv1, P.bn1, training)
    if FUSED_1X1 and ops.bn_relu_conv1x1_supported(r.dtype, D):
        a1 = None                                    # relu(bn1(h1)) is built inside the GEMM's operand staging, here and in the backward
        if training:    # bn2's batch statistics come out of the GEMM's store phase
            h2, m2, i2 = ops.bn_relu_conv1x1_forward_bnstats(h1, m1, i1, P.bn1.weight, P.bn1.bias, P.conv2.weight, P.conv2.bias,
                                                             P.bn2.running_mean, P.bn2.running_var)
            _bump(P.bn2)
        else:
            h2 = ops.bn_relu_conv1x1_forward(h1, m1, i1, P.bn1.weight, P.bn1.bias, P.conv2.weight, P.conv2.bias)
            m2, i2 = _bn_forward(h2, P.bn2, False)
    else:
        a1 = ops.bn_apply(h1, m1, i1, P.bn1.weight, P.bn1.bias, relu=True)
        h2, m2, i2 = _conv_bn(d2, a1, wf2, P.conv2, P.bn2, training)
    y = ops.bn_apply(h2, m2, i2, P.bn2.weight, P.bn2.bias, relu=False, residual=r, relu_residual=False, out_dtype=out_dtype,
                     relu_out=relu_out)
    saved = (r, h1, a1, h2, m1, i1, m2, i2, d1, d2, wd1, wd2)
    return y, saved


def resblock_bn2(saved):
    """(h2, mean, invstd) of a ResBlock's closing BatchNorm from its saved forward state, for a producer of the block's incoming
    gradient that forms that BatchNorm's backward sums itself (ops.vq_losses_indexed(bn=)) and passes them on as bn2_sums= --
    or None when the block's backward would not take them (it ran the separate operators: their BatchNorm backward is one call)."""
    r, h1, a1, h2, m1, i1, m2, i2 = saved[:8]
    return (h2, m2, i2) if a1 is None else None


def encoder_closing_bn(saved):
    """resblock_bn2 of the encoder's last ResBlock (the BatchNorm whose output is z_e), from encoder_forward's saved state."""
    return resblock_bn2(saved[9])


def resblock_backward(dy, saved, P: ResBlockParams, need_dx: bool = True, gout=None, bn2_sums=None):
    """Returns (dx, grads) with grads in the order conv1.w, conv1.b, bn1.w, bn1.b, conv2.w, conv2.b, bn2.w, bn2.b.
    gout: optional list of 8 preallocated tensors (e.g. views of a flat gradient bucket) to write into.
    bn2_sums: (dgamma, dbeta) of the closing BatchNorm when whoever produced dy has formed them already (written into gout[6],
    gout[7] if gout is given)."""
    x, h1, a1, h2, m1, i1, m2, i2, d1, d2, wd1, wd2 = saved
    o = gout if gout is not None else [None] * 8
    D = h2.shape[-1]
    # the conv biases sit in front of a BatchNorm: their gradient is the column sum of that
    # BatchNorm's input gradient, emitted by the BN-backward kernel itself (dx_colsum)
    dbias2 = o[5] if o[5] is not None else torch.empty(D, dtype=torch.float32, device=h2.device)
    dbias1 = o[1] if o[1] is not None else torch.empty(D, dtype=torch.float32, device=h2.device)
    if a1 is None:      # flat-GEMM 1x1: bn2's sums, then its apply + the conv's data gradient in one pass, the weight gradient from h1
        if bn2_sums is not None:
            dg2, db2n = bn2_sums
        else:
            dg2, db2n = ops.bn_backward_sums(h2, dy, m2, i2, P.bn2.weight, dgamma=o[6], dbeta=o[7])
        if FUSED_1X1_BWD and ops.bn_backward_conv1x1_dgrad_wgrad_supported(h2.dtype, D):     # data + weight gradient in one pass, dh2 never stored
            da1, dw2, dg1, db1n = ops.bn_backward_conv1x1_dgrad_wgrad(h2, dy, m2, i2, P.bn2.weight, dg2, db2n, P.conv2.weight,
                                                                      (h1, m1, i1, P.bn1.weight, P.bn1.bias), dh_colsum=dbias2, dw=o[4],
                                                                      prev_dgamma=o[2], prev_dbeta=o[3])
        else:
            dh2, da1, dg1, db1n = ops.bn_backward_conv1x1_dgrad(h2, dy, m2, i2, P.bn2.weight, dg2, db2n, P.conv2.weight, dh_colsum=dbias2,
                                                                prev=(h1, m1, i1, P.bn1.weight, P.bn1.bias), prev_dgamma=o[2], prev_dbeta=o[3])
            dw2 = ops.bn_relu_conv1x1_wgrad(h1, m1, i1, P.bn1.weight, P.bn1.bias, dh2, dw=o[4])
        dh1 = ops.bn_backward_apply(h1, da1, m1, i1, P.bn1.weight, dg1, db1n, relu_beta=P.bn1.bias, dx_colsum=dbias1)   # bn1's sums came with da1
    else:
        dh2, dg2, db2n = ops.bn_backward(h2, None, dy, m2, i2, P.bn2.weight, dgamma=o[6], dbeta=o[7], dx_colsum=dbias2)
        dw2, _ = ops.conv_wgrad(d2, a1, dh2, P.conv2.weight.shape, dw=o[4], want_bias=False)
        da1 = ops.conv_dgrad(d2, dh2, wd2)
    if a1 is not None:
        dh1, dg1, db1n = ops.bn_backward(h1, None, da1, m1, i1, P.bn1.weight, dgamma=o[2], dbeta=o[3], dx_colsum=dbias1,
                                         relu_beta=P.bn1.bias)   # ReLU mask re-derived from h1: a1 is not read
    dw1, _ = ops.conv_wgrad(d1, x, dh1, P.conv1.weight.shape, dw=o[0], want_bias=False)   # x is the stored relu(x)
    dx = None
    if need_dx:
        dx = ops.conv_dgrad(d1, dh1, wd1, add=dy, relu_x=x)    # (dgrad + skip-path gradient) * (x > 0), one kernel
    return dx, [dw1, dbias1, dg1, db1n, dw2, dbias2, dg2, db2n]


# ------------------------------------------------------------------------------------------------
# Encoder   (src/models.py:164-171)
# ------------------------------------------------------------------------------------------------
def encoder_forward(x, P: EncoderParams, training: bool, dtype=torch.float32, packs=None):
    """x fp32 NHWC (B, H, W, 1) -> z_e fp32 NHWC (B, H/4, W/4, D).  dtype: storage type of the activations in
    between (fp32 = parity mode, bf16 = throughput mode); the quantiser input z_e is fp32 in both."""
    B, H, W, _ = x.shape
    D = P.conv0.weight.shape[0]
    d0 = ops.conv_desc(B, H, W, 1, D, 4, 2, 1, dtype=dtype)
    pk = packs if packs is not None else {}
    wf0, _ = pk["conv0"] if packs is not None else ops.pack_weights(d0, P.conv0.weight, want_dgrad=False)
    mom0 = None
    if FUSED_C1_LAYER and D % 4 == 0 and D <= 1024 and H % 2 == 0 and W % 2 == 0:
        h0 = None                                    # never materialised; the backward recomputes it from x
        if training:
            mom0 = torch.empty(ops.C1_MOMENTS, dtype=torch.float64, device=x.device)     # the image's tap moments: forward statistics, backward
            a0, m0, i0 = ops.c1conv_bn_relu_forward(x, P.conv0.weight, P.conv0.bias, P.bn0.weight, P.bn0.bias, P.bn0.running_mean,
                                                    P.bn0.running_var, training=True, out_dtype=dtype, moments=mom0)
            _bump(P.bn0)
        else:
            m0, i0 = ops.bn_eval_stats(P.bn0.running_mean, P.bn0.running_var)
            a0, _, _ = ops.c1conv_bn_relu_forward(x, P.conv0.weight, P.conv0.bias, P.bn0.weight, P.bn0.bias, training=False,
                                                  mean=m0, invstd=i0, out_dtype=dtype)
    else:
        h0, m0, i0 = _conv_bn(d0, x, wf0, P.conv0, P.bn0, training)
        a0 = ops.bn_apply(h0, m0, i0, P.bn0.weight, P.bn0.bias, relu=True)
    d3 = ops.conv_desc(B, d0.OH, d0.OW, D, D, 4, 2, 1, dtype=dtype)
    wf3, wd3 = pk["conv3"] if packs is not None else ops.pack_weights(d3, P.conv3.weight)
    e3 = ops.conv_forward(d3, a0, wf3, P.conv3.bias, flags=NSG_RELU_OUT)     # stored ReLU'd: its only consumer is a ResBlock
    r4, s4 = resblock_forward(e3, P.res4, training, relu_out=True, packs=pk.get("res4"))
    ze, s5 = resblock_forward(r4, P.res5, training, out_dtype=torch.float32, packs=pk.get("res5"))
    saved = (x, h0, a0, m0, i0, d0, d3, wd3, s4, s5, mom0)
    return ze, saved


def encoder_backward(dze, saved, P: EncoderParams, gout=None, bn2_sums=None):
    """Gradients of every encoder parameter, in state_dict order (input gets none: it is data).
    dze must have the encoder's compute dtype.  gout: optional list of 22 preallocated tensors to write into.
    bn2_sums: (dgamma, dbeta) of the last ResBlock's closing BatchNorm, when dze's producer formed them (resblock_backward)."""
    x, h0, a0, m0, i0, d0, d3, wd3, s4, s5, mom0 = saved
    dze = ops.convert(dze, a0.dtype)
    o = gout if gout is not None else [None] * 22
    dr4, g5 = resblock_backward(dze, s5, P.res5, gout=o[14:22] if gout is not None else None, bn2_sums=bn2_sums)
    de3, g4 = resblock_backward(dr4, s4, P.res4, gout=o[6:14] if gout is not None else None)
    dw3, db3 = ops.conv_wgrad(d3, a0, de3, P.conv3.weight.shape, dw=o[4], dbias=o[5])
    da0 = ops.conv_dgrad(d3, de3, wd3)
    if h0 is None:      # fused input layer: BatchNorm backward and the weight gradient straight from (x, da0)
        dw0, db0, dg0, dbe0 = ops.c1conv_bn_relu_backward(x, P.conv0.weight, P.conv0.bias, P.bn0.weight, P.bn0.bias, m0, i0, da0,
                                                          dw=o[0], dbias=o[1], dgamma=o[2], dbeta=o[3], moments=mom0)
        return [dw0, db0, dg0, dbe0, dw3, db3] + g4 + g5
    db0 = o[1] if o[1] is not None else torch.empty(h0.shape[-1], dtype=torch.float32, device=h0.device)
    dh0, dg0, dbe0 = ops.bn_backward(h0, None, da0, m0, i0, P.bn0.weight, dgamma=o[2], dbeta=o[3], dx_colsum=db0, relu_beta=P.bn0.bias)
    dw0, _ = ops.conv_wgrad(d0, x, dh0, P.conv0.weight.shape, dw=o[0], want_bias=False)
    return [dw0, db0, dg0, dbe0, dw3, db3] + g4 + g5


# ------------------------------------------------------------------------------------------------
# Decoder   (src/models.py:175-184)
# ------------------------------------------------------------------------------------------------
def decoder_forward(zq, P: DecoderParams, training: bool, dtype=torch.float32, packs=None, zq_is_relu=False, mse_target=None,
                    mse_dbias=None):
    """zq NHWC (B, h, w, D) -> x_tilde fp32 NHWC (B, 4h, 4w, 1); activations in between stored as dtype.
    zq_is_relu: zq already holds max(0, z_q) in `dtype` (the quantiser wrote it: ops.vq_forward codes_bf16="relu").
    mse_target: the training step's reconstruction target (B, 4h, T, 1) fp32.  Where the fused output layer runs, the loss and
    the gradient w.r.t. the Tanh's input come out of the pass that forms the image: returns ((loss, dpre), saved) instead of
    (x_tilde, saved) -- test with isinstance(result, tuple); decoder_backward then takes dpre with dxt_is_pre_tanh=True.
    mse_dbias: optional (1,) tensor for the output conv's bias gradient (= sum of dpre), formed in the same pass; the backward
    then leaves it alone."""
    B, H, W, D = zq.shape
    if not (zq_is_relu and zq.dtype == dtype):
        zq = ops.convert(zq, dtype, relu=True)                   # decoder.0's leading ReLU, applied once here
    pk = packs if packs is not None else {}
    r0, s0 = resblock_forward(zq, P.res0, training, relu_out=True, packs=pk.get("res0"))
    r1, s1 = resblock_forward(r0, P.res1, training, relu_out=True, packs=pk.get("res1"))   # decoder.2 ReLU applied at the producer
    dT = ops.conv_desc(B, H, W, D, D, 4, 2, 1, transposed=True, dtype=dtype)
    wfT, wdT = pk["convt3"] if packs is not None else ops.pack_weights(dT, P.convt3.weight)
    u, m, i = _conv_bn(dT, r1, wfT, P.convt3, P.bn4, training)
    d6 = ops.conv_desc(B, dT.OH, dT.OW, D, 1, 4, 2, 1, transposed=True, dtype=dtype)
    if FUSED_OUT_LAYER and ops.bn_relu_c1convt_supported(u.dtype, D):
        a, wd6 = None, None                          # relu(bn(u)) is never materialised; the backward rebuilds it from u
        if mse_target is not None and FUSED_OUT_LOSS:
            loss, dpre, _ = ops.bn_relu_c1convt_forward_mse(u, m, i, P.bn4.weight, P.bn4.bias, P.convt6.weight, P.convt6.bias, mse_target,
                                                            dbias=mse_dbias)
            return (loss, dpre), (r1, u, a, m, i, ("bias gradient done", mse_dbias) if mse_dbias is not None else None, dT, d6, wdT, wd6, s0, s1)
        xt = ops.bn_relu_c1convt_forward(u, m, i, P.bn4.weight, P.bn4.bias, P.convt6.weight, P.convt6.bias, tanh=True)
    else:
        a = ops.bn_apply(u, m, i, P.bn4.weight, P.bn4.bias, relu=True)
        wf6, wd6 = pk["convt6"] if packs is not None else ops.pack_weights(d6, P.convt6.weight)
        xt = ops.conv_forward(d6, a, wf6, P.convt6.bias, flags=NSG_TANH_OUT)  # decoder.7 Tanh fused into the epilogue
    saved = (r1, u, a, m, i, xt, dT, d6, wdT, wd6, s0, s1)
    return xt, saved


def decoder_backward(dxt, saved, P: DecoderParams, need_dz: bool = True, dxt_is_pre_tanh: bool = False, gout=None):
    """dxt: gradient w.r.t. x_tilde (or w.r.t. the tanh input when dxt_is_pre_tanh).
    Returns (dzq, grads in state_dict order).  gout: optional list of 22 preallocated tensors."""
    r1, u, a, m, i, xt, dT, d6, wdT, wd6, s0, s1 = saved
    o = gout if gout is not None else [None] * 22
    dpre = dxt if dxt_is_pre_tanh else ops.tanh_backward(dxt, xt)
    dbT = o[17] if o[17] is not None else torch.empty(u.shape[-1], dtype=torch.float32, device=u.device)
    if a is None:       # fused output layer: BatchNorm backward, the transposed conv's data and weight gradients in two passes over u
        db_done = xt[1] if isinstance(xt, tuple) else None       # the forward's loss pass already summed dpre (decoder_forward: mse_dbias)
        du, dw6, db6, dg4, dbe4 = ops.bn_relu_c1convt_backward(u, m, i, P.bn4.weight, P.bn4.bias, P.convt6.weight, dpre, dw=o[20],
                                                               dbias=db_done if db_done is not None else o[21], dgamma=o[18], dbeta=o[19],
                                                               du_colsum=dbT, want_dbias=db_done is None)
    else:
        dw6, db6 = ops.conv_wgrad(d6, a, dpre, P.convt6.weight.shape, dw=o[20], dbias=o[21])
        da = ops.conv_dgrad(d6, dpre, wd6)
        du, dg4, dbe4 = ops.bn_backward(u, None, da, m, i, P.bn4.weight, dgamma=o[18], dbeta=o[19], dx_colsum=dbT, relu_beta=P.bn4.bias)
    dwT, _ = ops.conv_wgrad(dT, r1, du, P.convt3.weight.shape, dw=o[16], want_bias=False)   # r1 is stored ReLU'd
    dr1 = ops.conv_dgrad(dT, du, wdT, relu_x=r1)               # decoder.2 ReLU's mask applied in the dgrad store
    dr0, g1 = resblock_backward(dr1, s1, P.res1, gout=o[8:16] if gout is not None else None)
    dzq, g0 = resblock_backward(dr0, s0, P.res0, need_dx=need_dz, gout=o[0:8] if gout is not None else None)
    return dzq, g0 + g1 + [dwT, dbT, dg4, dbe4, dw6, db6]


def pack_all(encP: EncoderParams, decP: DecoderParams, B: int, H: int, W: int, dtype):
    """Every packed weight image of one training step in ONE launch (nsg_pack_conv_weights_batch).
    (B, H, W): the mel image extent.  Returns (encoder packs, decoder packs) for encoder_forward / decoder_forward."""
    D = encP.conv0.weight.shape[0]
    d0 = ops.conv_desc(B, H, W, 1, D, 4, 2, 1, dtype=dtype)
    d3 = ops.conv_desc(B, d0.OH, d0.OW, D, D, 4, 2, 1, dtype=dtype)
    h, w = d3.OH, d3.OW
    r3 = ops.conv_desc(B, h, w, D, D, 3, 1, 1, dtype=dtype)
    r1 = ops.conv_desc(B, h, w, D, D, 1, 1, 0, dtype=dtype)
    dT = ops.conv_desc(B, h, w, D, D, 4, 2, 1, transposed=True, dtype=dtype)
    d6 = ops.conv_desc(B, dT.OH, dT.OW, D, 1, 4, 2, 1, transposed=True, dtype=dtype)
    jobs = [(d0, encP.conv0.weight, True, False), (d3, encP.conv3.weight, True, True)]
    for rb in (encP.res4, encP.res5, decP.res0, decP.res1):
        jobs += [(r3, rb.conv1.weight, True, True), (r1, rb.conv2.weight, True, True)]
    jobs += [(dT, decP.convt3.weight, True, True), (d6, decP.convt6.weight, True, True)]
    pk = ops.pack_weights_batch(jobs)
    enc = {"conv0": pk[0], "conv3": pk[1], "res4": (pk[2], pk[3]), "res5": (pk[4], pk[5])}
    dec = {"res0": (pk[6], pk[7]), "res1": (pk[8], pk[9]), "convt3": pk[10], "convt6": pk[11]}
    return enc, dec


# ------------------------------------------------------------------------------------------------
# helpers to pull parameter bundles out of the nn.Module tree (names follow the reference)
# ------------------------------------------------------------------------------------------------
def conv_params(m) -> ConvParams:
    return ConvParams(m.weight, m.bias)


def bn_params(m) -> BNParams:
    return BNParams(m.weight, m.bias, m.running_mean, m.running_var, m.num_batches_tracked)


def resblock_params(rb) -> ResBlockParams:
    return ResBlockParams(conv_params(rb.block[1]), bn_params(rb.block[2]), conv_params(rb.block[4]), bn_params(rb.block[5]))


def encoder_params(enc) -> EncoderParams:
    return EncoderParams(conv_params(enc[0]), bn_params(enc[1]), conv_params(enc[3]), resblock_params(enc[4]),
                         resblock_params(enc[5]))


def decoder_params(dec) -> DecoderParams:
    return DecoderParams(resblock_params(dec[0]), resblock_params(dec[1]), conv_params(dec[3]), bn_params(dec[4]),
                         conv_params(dec[6]))


def resblock_param_list(P: ResBlockParams) -> List[torch.Tensor]:
    return [P.conv1.weight, P.conv1.bias, P.bn1.weight, P.bn1.bias, P.conv2.weight, P.conv2.bias, P.bn2.weight, P.bn2.bias]


def encoder_param_list(P: EncoderParams) -> List[torch.Tensor]:
    return [P.conv0.weight, P.conv0.bias, P.bn0.weight, P.bn0.bias, P.conv3.weight, P.conv3.bias] + \
        resblock_param_list(P.res4) + resblock_param_list(P.res5)


def decoder_param_list(P: DecoderParams) -> List[torch.Tensor]:
    return resblock_param_list(P.res0) + resblock_param_list(P.res1) + \
        [P.convt3.weight, P.convt3.bias, P.bn4.weight, P.bn4.bias, P.convt6.weight, P.convt6.bias]
