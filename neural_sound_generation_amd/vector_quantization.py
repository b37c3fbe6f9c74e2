"""Operator surface of the reference's src/vector_quantization.py, on hand-written HIP kernels.

    vq(inputs[..., D], codebook[K, D])     -> int64 indices[...]           (not differentiable)
    vq_st(inputs[..., D], codebook[K, D])  -> (codes like inputs, indices_flat[N])   straight-through

Same names, argument meaning and error behaviour as the reference (vector_quantization.py:4-66):
differentiating through plain `vq` raises RuntimeError (:26-30); `vq_st` passes the output gradient
straight to the inputs (:52) and scatter-adds it into the codebook gradient when the codebook
requires grad (:53-61).  The search itself is nsg_vq_forward: distances are never materialised and
the indices are bit-identical to the reference's CPU result.  The optional third argument
`impl` ("mfma" by default) lets the bf16 compute mode select its own search ("bf16x3", nsg_vq_forward_bf16x3).
"""
from __future__ import annotations

import torch
from torch.autograd import Function

from . import ops


class VectorQuantization(Function):
    @staticmethod
    def forward(ctx, inputs, codebook, impl="mfma"):
        with torch.no_grad():
            embedding_size = codebook.size(1)
            flat = inputs.detach().contiguous().view(-1, embedding_size)
            idx, _, _ = ops.vq_forward(flat, codebook.detach().contiguous(), want_codes=False, impl=impl)
            indices = idx.view(*inputs.shape[:-1])
            ctx.mark_non_differentiable(indices)
            return indices

    @staticmethod
    def backward(ctx, grad_output):
        raise RuntimeError('Trying to call `.grad()` on graph containing `VectorQuantization`. '
                           'The function `VectorQuantization` is not differentiable. '
                           'Use `VectorQuantizationStraightThrough` if you want a straight-through '
                           'estimator of the gradient.')


class VectorQuantizationStraightThrough(Function):
    @staticmethod
    def forward(ctx, inputs, codebook, impl="mfma"):
        embedding_size = codebook.size(1)
        flat = inputs.detach().contiguous().view(-1, embedding_size)
        cb = codebook.detach().contiguous()
        indices_flatten, codes_flatten, _ = ops.vq_forward(flat, cb, want_codes=True, impl=impl)  # search + gather fused
        ctx.save_for_backward(indices_flatten)
        ctx.codebook_rows = cb.size(0)
        ctx.mark_non_differentiable(indices_flatten)
        return codes_flatten.view_as(inputs), indices_flatten

    @staticmethod
    def backward(ctx, grad_output, grad_indices):
        grad_inputs, grad_codebook = None, None
        if ctx.needs_input_grad[0]:
            grad_inputs = ops.add(grad_output.contiguous(), None)  # straight-through estimator (a copy)
        if ctx.needs_input_grad[1]:
            (indices,) = ctx.saved_tensors
            g = grad_output.contiguous().view(indices.numel(), -1)
            grad_codebook = ops.index_add_rows(indices, g, ctx.codebook_rows)
        return grad_inputs, grad_codebook, None


class CodebookLookup(Function):
    """codebook[indices] with gradient to the codebook: the reference's
    torch.index_select(self.embedding.weight, 0, indices) (src/models.py:137-138)."""

    @staticmethod
    def forward(ctx, codebook, indices, impl="f32"):
        ctx.save_for_backward(indices)
        ctx.codebook_rows = codebook.size(0)
        ctx.impl = impl
        return ops.gather_rows(codebook.detach().contiguous(), indices)

    @staticmethod
    def backward(ctx, grad_output):
        (indices,) = ctx.saved_tensors
        g = grad_output.contiguous().view(indices.numel(), -1)
        return ops.index_add_rows(indices.view(-1), g, ctx.codebook_rows, impl=ctx.impl), None, None


class AddPerClip(Function):
    """z (B,H,W,D) + rows (B,D) broadcast over pixels -- the speaker-conditioning add (extension)."""

    @staticmethod
    def forward(ctx, z, rows):
        ctx.batch = rows.shape[0]
        return ops.add_per_clip(z.detach().contiguous(), rows.detach().contiguous())

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        gz = ops.add(g, None) if ctx.needs_input_grad[0] else None
        gr = ops.clip_colsum(g, ctx.batch) if ctx.needs_input_grad[1] else None
        return gz, gr


def vq(inputs, codebook, impl="mfma"):
    return VectorQuantization.apply(inputs, codebook, impl)


def vq_st(inputs, codebook, impl="mfma"):
    return VectorQuantizationStraightThrough.apply(inputs, codebook, impl)


def codebook_lookup(codebook, indices, impl="f32"):
    return CodebookLookup.apply(codebook, indices, impl)


add_per_clip = AddPerClip.apply
__all__ = ["vq", "vq_st", "codebook_lookup"]
