"""torch.autograd.Function wrappers around the engine's explicit forward/backward pieces, so the
fused encoder / decoder / ResBlock stacks compose with ordinary autograd code (the reference's
src/train.py calls loss.backward() on the model outputs).

Tensors cross this boundary in the reference's logical NCHW shape; physically they are NHWC
("channels_last"), so the permutes below are views, not copies.
"""
from __future__ import annotations

import torch
from torch.autograd import Function

from . import engine, ops


def to_nhwc(x: torch.Tensor) -> torch.Tensor:
    """(B,C,H,W) logical -> contiguous [B][H][W][C]; free when x is channels_last or C == 1."""
    return x.permute(0, 2, 3, 1).contiguous()


def to_nchw_view(x: torch.Tensor) -> torch.Tensor:
    return x.permute(0, 3, 1, 2)


def _check_float_cuda(x, what):
    if not x.is_cuda:
        raise RuntimeError(f"{what}: this path runs only on an AMD GPU (got a {x.device} tensor); there is no CPU fallback")
    if x.dtype != torch.float32:
        raise RuntimeError(f"{what}: expected float32, got {x.dtype}")


class _Stack(Function):
    """Shared plumbing: subclasses set fwd / bwd / bundle."""

    @staticmethod
    def _run_forward(ctx, fwd, bundle, x, training, params, dtype):
        _check_float_cuda(x, fwd.__name__)
        ctx.bundle = bundle
        ctx.training = training
        ctx.dtype = dtype
        # The backward reads some parameters again (BatchNorm scales, the 1x1 weights): registering them with autograd makes
        # `ctx.saved_tensors` raise the usual "modified by an inplace operation" error if an optimiser stepped in between,
        # as the reference's per-layer autograd would, instead of silently differentiating against the new values.
        ctx.save_for_backward(*params)
        if fwd is engine.resblock_forward:
            y, saved = fwd(ops.convert(to_nhwc(x.detach()), dtype, relu=True), bundle, training, out_dtype=torch.float32)
        else:
            y, saved = fwd(to_nhwc(x.detach()), bundle, training, dtype=dtype)
        ctx.saved = saved
        return to_nchw_view(y)


class EncoderFn(Function):
    @staticmethod
    def forward(ctx, x, bundle, training, dtype, *params):
        if ctx.needs_input_grad[0]:
            # the reference can differentiate w.r.t. the mel image; this path does not build that gradient (the input layer's
            # conv output is never stored) -- say so instead of returning None
            raise RuntimeError("the encoder input is data: a gradient w.r.t. the mel batch is not implemented (pass c.detach())")
        return _Stack._run_forward(ctx, engine.encoder_forward, bundle, x, training, params, dtype)

    @staticmethod
    def backward(ctx, dz):
        if not ctx.training:
            raise RuntimeError("backward through the encoder in eval() mode is not implemented")
        ctx.saved_tensors                      # version check of the parameters (see _run_forward)
        grads = engine.encoder_backward(to_nhwc(dz), ctx.saved, ctx.bundle)
        return (None, None, None, None) + tuple(grads)


class DecoderFn(Function):
    @staticmethod
    def forward(ctx, z, bundle, training, dtype, *params):
        return _Stack._run_forward(ctx, engine.decoder_forward, bundle, z, training, params, dtype)

    @staticmethod
    def backward(ctx, dxt):
        if not ctx.training:
            raise RuntimeError("backward through the decoder in eval() mode is not implemented")
        ctx.saved_tensors
        dz, grads = engine.decoder_backward(to_nhwc(dxt), ctx.saved, ctx.bundle, need_dz=ctx.needs_input_grad[0])
        if dz is not None:
            dz = to_nchw_view(ops.convert(dz, torch.float32))
        return (dz, None, None, None) + tuple(grads)


class ResBlockFn(Function):
    @staticmethod
    def forward(ctx, x, bundle, training, dtype, *params):
        return _Stack._run_forward(ctx, engine.resblock_forward, bundle, x, training, params, dtype)

    @staticmethod
    def backward(ctx, dy):
        if not ctx.training:
            raise RuntimeError("backward through a ResBlock in eval() mode is not implemented")
        ctx.saved_tensors
        dx, grads = engine.resblock_backward(ops.convert(to_nhwc(dy), ctx.dtype), ctx.saved, ctx.bundle, need_dx=ctx.needs_input_grad[0])
        if dx is not None:
            dx = to_nchw_view(ops.convert(dx, torch.float32))
        return (dx, None, None, None) + tuple(grads)


def encoder_apply(x, bundle, training, dtype=torch.float32):
    return EncoderFn.apply(x, bundle, training, dtype, *engine.encoder_param_list(bundle))


def decoder_apply(z, bundle, training, dtype=torch.float32):
    return DecoderFn.apply(z, bundle, training, dtype, *engine.decoder_param_list(bundle))


def resblock_apply(x, bundle, training, dtype=torch.float32):
    return ResBlockFn.apply(x, bundle, training, dtype, *engine.resblock_param_list(bundle))
