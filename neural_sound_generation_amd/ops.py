"""Tensor-level wrappers over the C ABI: validate, allocate outputs/workspaces as torch tensors
(so the caching allocator and stream semantics apply), pass raw pointers + the current HIP stream.
PyTorch is plumbing here: device memory, streams, autograd glue -- all arithmetic is in libnsg.so.
All activations are fp32 NHWC; see include/nsg.h.
"""
from __future__ import annotations

import ctypes
from ctypes import byref, c_float, c_int32, c_int64, c_size_t, c_void_p

import torch

from . import _lib
from ._lib import ConvDesc, NSG_RELU_IN, NSG_TANH_OUT, NSG_OUT_F32, NSG_RELU_OUT, NSG_F32, NSG_BF16  # noqa: F401

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def _stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)


def nsg_dtype(torch_dtype) -> int:
    if torch_dtype == torch.float32:
        return NSG_F32
    if torch_dtype == torch.bfloat16:
        return NSG_BF16
    raise _lib.NsgError(f"unsupported activation dtype {torch_dtype} (float32 or bfloat16)")


def torch_dtype(nsg: int):
    return torch.bfloat16 if nsg == NSG_BF16 else torch.float32


def _chk(t, name, dtype=torch.float32):
    if not t.is_cuda:
        raise _lib.NsgError(f"{name}: expected a GPU tensor (this path has no CPU fallback)")
    if dtype is not None and t.dtype != dtype:
        raise _lib.NsgError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise _lib.NsgError(f"{name}: expected a contiguous tensor")
    return t


class Workspace:
    """One growing scratch buffer per (device, stream): kernels on one stream run in order, so reuse is safe; a second stream
    (engine.py runs the weight gradients on one) gets a buffer of its own."""

    def __init__(self):
        self._buf = {}

    @staticmethod
    def _key(device):
        sid = torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0
        return (device.type, device.index, sid)

    def get(self, nbytes: int, device) -> torch.Tensor:
        key = self._key(device)
        buf = self._buf.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
            self._buf[key] = buf
        return buf

    def current(self, device):
        """The buffer get() hands out for `device` on the current stream right now (None before first use).  Growth REPLACES
        it; whoever baked its address into a HIP graph keeps a reference to this tensor so the address stays theirs
        (FusedTrainStep.capture)."""
        return self._buf.get(self._key(device))


WS = Workspace()


class KernelTimer:
    """Optional HIP-event timing of the single-kernel GEMM launches (bench.py's roofline figure).
    Events are recorded on the stream the kernels are launched on (torch's current stream)."""

    def __init__(self):
        self.records = {}   # name -> list of (start_event, end_event, flops)

    def begin(self):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        return ev

    def end(self, name, start, flops, nbytes=0.0):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        self.records.setdefault(name, []).append((start, ev, flops, nbytes))

    def summary(self):
        """name -> dict(launches, total_ms, avg_ms, flops_per_launch, bytes_per_launch, tflops, tbps)  (call after a synchronize)"""
        out = {}
        for name, recs in self.records.items():
            ms = [r[0].elapsed_time(r[1]) for r in recs]
            tot_ms, tot_fl, tot_b = sum(ms), float(sum(r[2] for r in recs)), float(sum(r[3] for r in recs))
            out[name] = dict(launches=len(recs), total_ms=tot_ms, avg_ms=tot_ms / len(recs),
                             flops_per_launch=tot_fl / len(recs), bytes_per_launch=tot_b / len(recs),
                             tflops=tot_fl / (tot_ms * 1e-3) / 1e12 if tot_ms > 0 else 0.0,
                             tbps=tot_b / (tot_ms * 1e-3) / 1e12 if tot_ms > 0 else 0.0)
        return out


KERNEL_TIMER = None  # set to a KernelTimer() to time gather_gemm / wgrad launches


def _es(t) -> int:
    return t.element_size()


def _conv_label(d: "ConvDesc", role: str) -> str:
    """Kernel + geometry label for the census: which gather_gemm mode a forward / dgrad of this layer runs (a transposed conv's
    forward and a stride-2 conv's data gradient are the 4-parity-class form, MODE 1; everything else is the conv gather, MODE 0)."""
    kw = d.k_w if d.k_w > 0 else d.k
    geom = f"{d.k}x{kw}" + (f"/s{d.stride}" if d.stride != 1 else "")
    if role == "wgrad":
        return f"wgrad_gemm {geom}{' transposed' if d.transposed else ''} {d.C_in}->{d.C_out}"
    mode1 = bool(d.transposed) if role == "forward" else (not d.transposed and d.stride == 2)
    return f"gather_gemm MODE {1 if mode1 else 0} {geom} {role} {d.C_in}->{d.C_out}"


def _gemm_flops(d: "ConvDesc") -> float:
    """Algorithmic FLOPs of one forward / dgrad / wgrad of layer d: 2 * low-res pixels * k^2 * C_in * C_out."""
    low = d.B * (d.IH * d.IW if d.transposed else d.OH * d.OW)
    return 2.0 * low * d.k * (d.k_w if d.k_w > 0 else d.k) * d.C_in * d.C_out


# ------------------------------------------------------------------------------------------------
# vector quantiser
# ------------------------------------------------------------------------------------------------
def vq_forward(x2d, codebook, want_codes=True, want_dist=False, impl="mfma", codes_bf16=None, clip_rows=None):
    """x2d (N,D), codebook (K,D) -> idx (N,) int64 [, codes (N,D)] [, dmin (N,)]
    impl: "mfma" = the bit-exact fp32 search (parity mode); "valu" = its vector-ALU cross-check; "bf16x3" = the bf16
    mode's search on the bf16 matrix pipe with split operands (relative distance error ~2^-16: near-ties may differ).
    codes_bf16 ("plain" | "relu", bf16x3 only): also return the code rows as a bf16 (N,D) tensor (ReLU'd: the decoder's
    input after its leading ReLU) as a 4th result.  clip_rows (B, D) fp32 (with codes_bf16): a per-clip conditioning row added
    to every bf16 code row of that clip before the ReLU (N = B * rows per clip; the speaker-conditioned decoder)."""
    _chk(x2d, "x"); _chk(codebook, "codebook")
    if clip_rows is not None:
        _chk(clip_rows, "clip_rows")
        if impl != "bf16x3" or not codes_bf16 or clip_rows.shape[1] != x2d.shape[1] or x2d.shape[0] % clip_rows.shape[0] != 0:
            raise ValueError("vq_forward: clip_rows needs impl='bf16x3', codes_bf16 and (B, D) rows with N a multiple of B")
    N, D = x2d.shape
    K, D2 = codebook.shape
    if D != D2:
        raise _lib.NsgError(f"vq_forward: input rows have {D} columns, codebook has {D2}")
    idx = torch.empty(N, dtype=torch.int64, device=x2d.device)
    codes = torch.empty_like(x2d) if want_codes else None
    dmin = torch.empty(N, dtype=torch.float32, device=x2d.device) if want_dist else None
    if N > 0:
        wsfn = "nsg_vq_bf16x3_workspace_bytes" if impl == "bf16x3" else "nsg_vq_workspace_bytes"
        nb = _lib.query(wsfn, c_int64(N), c_int32(D), c_int32(K))
        ws = WS.get(nb, x2d.device)
        fn = {"mfma": "nsg_vq_forward", "valu": "nsg_debug_vq_forward_valu", "bf16x3": "nsg_vq_forward_bf16x3"}[impl]
        if impl == "bf16x3":
            lp = torch.empty(N, D, dtype=torch.bfloat16, device=x2d.device) if codes_bf16 else None
            _lib.tag("vq_forward_bf16x3 (search + gather)", 2.0 * N * K * D)
            if clip_rows is not None:
                _lib.call("nsg_vq_forward_bf16x3_cond", _p(x2d), _p(codebook), c_int64(N), c_int32(D), c_int32(K), _p(idx), _p(codes), _p(dmin),
                          _p(lp), c_int32(1 if codes_bf16 == "relu" else 0), _p(clip_rows), c_int64(N // clip_rows.shape[0]), _p(ws),
                          c_size_t(nb), _stream())
            else:
                _lib.call(fn, _p(x2d), _p(codebook), c_int64(N), c_int32(D), c_int32(K), _p(idx), _p(codes), _p(dmin), _p(lp),
                          c_int32(1 if codes_bf16 == "relu" else 0), _p(ws), c_size_t(nb), _stream())
            if codes_bf16:
                return idx, codes, dmin, lp
            return idx, codes, dmin
        if codes_bf16:
            raise ValueError("vq_forward: codes_bf16 needs impl='bf16x3'")
        _lib.tag("vq_forward (fp32 exact search + gather)", 2.0 * N * K * D)
        _lib.call(fn, _p(x2d), _p(codebook), c_int64(N), c_int32(D), c_int32(K), _p(idx), _p(codes), _p(dmin),
                  _p(ws), c_size_t(nb), _stream())
    elif codes_bf16:
        return idx, codes, dmin, torch.empty(0, D, dtype=torch.bfloat16, device=x2d.device)
    return idx, codes, dmin


def rowsumsq(v):
    _chk(v, "v")
    out = torch.empty(v.shape[0], dtype=torch.float32, device=v.device)
    _lib.call("nsg_rowsumsq", _p(v), c_int64(v.shape[0]), c_int32(v.shape[1]), _p(out), _stream())
    return out


def index_add_sorted_supported(N, D, K) -> bool:
    """Shapes nsg_index_add_rows_sorted takes (include/nsg.h)."""
    ppr = D // 4
    return D % 4 == 0 and K <= 8192 and N < 2 ** 31 and (ppr > 64 and D % 256 == 0 or ppr <= 64 and ppr & (ppr - 1) == 0)


def index_add_rows(idx, g2d, K, want_counts=False, impl="f32", out=None, counts=None):
    """out[k] = sum of rows of g2d whose idx == k; deterministic.  impl: "sorted" (a sorted segment sum in fp32: N*D*4 bytes
    moved; the training step's default), "f32" (one-hot GEMM on the fp32 matrix pipe, exact products) or "bf16x2" (one-hot GEMM
    on the bf16 pipe: rows split into bf16 hi + lo, relative error of a sum ~2^-17).
    out (K, D) / counts (K,): optional preallocated fp32 destinations (e.g. views of a communication buffer)."""
    _chk(idx, "idx", torch.int64); _chk(g2d, "g")
    N, D = g2d.shape
    if out is None:
        out = torch.empty(K, D, dtype=torch.float32, device=g2d.device)
    elif _chk(out, "out").shape != (K, D):
        raise _lib.NsgError(f"index_add_rows: out has shape {tuple(out.shape)}, expected {(K, D)}")
    if counts is not None:
        if _chk(counts, "counts").shape != (K,):
            raise _lib.NsgError(f"index_add_rows: counts has shape {tuple(counts.shape)}, expected {(K,)}")
        want_counts = True
    elif want_counts:
        counts = torch.empty(K, dtype=torch.float32, device=g2d.device)
    if impl == "sorted" and not index_add_sorted_supported(N, D, K):
        impl = "f32"
    if impl == "sorted":
        if N == 0:
            out.zero_()
            if counts is not None:
                counts.zero_()
            return (out, counts) if want_counts else out
        nb = _lib.query("nsg_index_add_sorted_workspace_bytes", c_int64(N), c_int32(D), c_int32(K))
        ws = WS.get(nb, g2d.device)
        _lib.tag("index_add_rows (sorted segment sum)", 0, 4.0 * N * D + 8.0 * N)
        _lib.call("nsg_index_add_rows_sorted", _p(idx), _p(g2d), c_int64(N), c_int32(D), c_int32(K), _p(out), _p(counts), _p(ws), c_size_t(nb),
                  _stream())
        return (out, counts) if want_counts else out
    nb = _lib.query("nsg_index_add_workspace_bytes", c_int64(N), c_int32(D), c_int32(K))
    ws = WS.get(nb, g2d.device)
    _lib.tag("index_add_rows (one-hot GEMM, %s)" % impl, 2.0 * N * K * D, 4.0 * N * D + 8.0 * N)
    _lib.call("nsg_index_add_rows_bf16x2" if impl == "bf16x2" else "nsg_index_add_rows", _p(idx), _p(g2d), c_int64(N), c_int32(D),
              c_int32(K), _p(out), _p(counts), _p(ws), c_size_t(nb), _stream())
    return (out, counts) if want_counts else out


def codebook_grad_from_sums(codebook, n, s, scale, out):
    """out[k] = scale * (n[k] * codebook[k] - s[k]): the codebook gradient of mse(codebook[idx], sg(z)) from the per-code
    counts n (K,) and sums s (K, D) of the rows z assigned to each code (index_add_rows(..., want_counts=True))."""
    _chk(codebook, "codebook"); _chk(n, "n"); _chk(s, "s"); _chk(out, "out")
    K, D = codebook.shape
    if n.numel() != K or s.numel() != K * D or out.numel() != K * D:
        raise _lib.NsgError("codebook_grad_from_sums: n, s and out must match the codebook's (K, D)")
    _lib.call("nsg_codebook_grad_from_sums", _p(codebook), _p(n), _p(s), c_int32(K), c_int32(D), c_float(scale), _p(out), _stream())
    return out


def increment_counters(counters):
    """counters: int64 GPU scalars (BatchNorm2d.num_batches_tracked), each += 1, one launch."""
    n = len(counters)
    if n == 0:
        return
    arr = (c_void_p * n)()
    for i, t in enumerate(counters):
        if not t.is_cuda or t.dtype != torch.int64 or t.numel() != 1:
            raise _lib.NsgError("increment_counters: expected int64 GPU scalars")
        arr[i] = t.data_ptr()
    _lib.call("nsg_increment_counters", ctypes.cast(arr, c_void_p), c_int32(n), _stream())


def gather_rows(codebook, idx):
    """codebook (K,D), idx (...) int64 -> (..., D)"""
    _chk(codebook, "codebook"); _chk(idx, "idx", torch.int64)
    K, D = codebook.shape
    out = torch.empty(*idx.shape, D, dtype=torch.float32, device=codebook.device)
    _lib.call("nsg_gather_rows", _p(codebook), _p(idx), c_int64(idx.numel()), c_int32(D), c_int32(K), _p(out), _stream())
    return out


def vq_ema_update(codebook, ema_n, ema_s, n, s, decay=0.99, eps=1e-5):
    K, D = codebook.shape
    scratch = torch.empty(1, dtype=torch.float32, device=codebook.device)
    _lib.call("nsg_vq_ema_update", _p(codebook), _p(ema_n), _p(ema_s), _p(n), _p(s), c_int32(K), c_int32(D),
              c_float(decay), c_float(eps), _p(scratch), _stream())


def debug_dot(x, e, mode):
    out = torch.empty(x.shape[0], e.shape[0], dtype=torch.float32, device=x.device)
    _lib.call("nsg_debug_dot", _p(x), _p(e), c_int32(x.shape[0]), c_int32(x.shape[1]), c_int32(e.shape[0]), c_int32(mode),
              _p(out), _stream())
    return out


# ------------------------------------------------------------------------------------------------
# convolutions
# ------------------------------------------------------------------------------------------------
def conv_desc(B, IH, IW, C_in, C_out, k, stride, pad, transposed=False, dtype=torch.float32, out_hw=None) -> ConvDesc:
    """dtype: storage type of the layer's multi-channel activations and packed weights.  k and pad may be (rows, columns)
    pairs (a rectangular stride-1 Conv2d); out_hw crops a stride-1 output at the bottom / right."""
    kh, kw = (k if isinstance(k, (tuple, list)) else (k, k))
    ph, pw = (pad if isinstance(pad, (tuple, list)) else (pad, pad))
    if transposed:
        OH, OW = (IH - 1) * stride - 2 * ph + kh, (IW - 1) * stride - 2 * pw + kw
    else:
        OH, OW = (IH + 2 * ph - kh) // stride + 1, (IW + 2 * pw - kw) // stride + 1
    if out_hw is not None:
        OH, OW = out_hw
    rect = (kh != kw) or (ph != pw)
    return ConvDesc(B, IH, IW, C_in, OH, OW, C_out, kh, stride, ph, 1 if transposed else 0, nsg_dtype(dtype), kw if rect else 0, pw if rect else 0)


def _in_dtype(d: ConvDesc):
    """torch dtype of the layer INPUT tensor (single-channel images are always fp32)."""
    return torch.float32 if d.C_in == 1 else torch_dtype(d.dtype)


def _out_dtype(d: ConvDesc, flags=0):
    return torch.float32 if (d.C_out == 1 or (flags & NSG_OUT_F32)) else torch_dtype(d.dtype)


def pack_weights(d: ConvDesc, w, want_fwd=True, want_dgrad=True):
    _chk(w, "weight")
    n = _lib.query("nsg_packed_weight_floats", byref(d))
    wf = torch.empty(n, dtype=torch_dtype(d.dtype), device=w.device) if want_fwd else None
    wd = torch.empty(n, dtype=torch_dtype(d.dtype), device=w.device) if want_dgrad else None
    _lib.call("nsg_pack_conv_weights", byref(d), _p(w), _p(wf), _p(wd), _stream())
    return wf, wd


def pack_weights_batch(jobs):
    """jobs: list of (desc, weight, want_fwd, want_dgrad) -> list of (w_fwd, w_dgrad); ONE kernel launch for all of them
    (a training step re-packs every layer's weights after each optimiser step)."""
    n = len(jobs)
    if n == 0:
        return []
    descs = (ConvDesc * n)()
    wp, fp, dp = (c_void_p * n)(), (c_void_p * n)(), (c_void_p * n)()
    out = []
    for i, (d, w, want_fwd, want_dgrad) in enumerate(jobs):
        _chk(w, "weight")
        ne = _lib.query("nsg_packed_weight_floats", byref(d))
        wf = torch.empty(ne, dtype=torch_dtype(d.dtype), device=w.device) if want_fwd else None
        wd = torch.empty(ne, dtype=torch_dtype(d.dtype), device=w.device) if want_dgrad else None
        ctypes.memmove(ctypes.addressof(descs[i]), ctypes.addressof(d), ctypes.sizeof(ConvDesc))
        wp[i] = w.data_ptr()
        fp[i] = wf.data_ptr() if wf is not None else None
        dp[i] = wd.data_ptr() if wd is not None else None
        out.append((wf, wd))
    _lib.tag("pack_weights_batch", 0, sum(4.0 * j[1].numel() for j in jobs) + sum((a.numel() * _es(a) if a is not None else 0) + (b.numel() * _es(b) if b is not None else 0) for a, b in out))
    _lib.call("nsg_pack_conv_weights_batch", c_int32(n), ctypes.cast(descs, c_void_p), ctypes.cast(wp, c_void_p),
              ctypes.cast(fp, c_void_p), ctypes.cast(dp, c_void_p), _stream())
    return out


def _conv_ws(d, device):
    nb = _lib.query("nsg_conv_workspace_bytes", byref(d))
    return WS.get(nb, device), nb


def conv_forward(d: ConvDesc, x, w_fwd, bias, flags=0, out=None):
    """x NHWC (B,IH,IW,C_in) -> y NHWC (B,OH,OW,C_out)."""
    _chk(x, "x", _in_dtype(d))
    if tuple(x.shape) != (d.B, d.IH, d.IW, d.C_in):
        raise _lib.NsgError(f"conv_forward: input shape {tuple(x.shape)} does not match descriptor {d.key()}")
    y = out if out is not None else torch.empty(d.B, d.OH, d.OW, d.C_out, dtype=_out_dtype(d, flags), device=x.device)
    ws, nb = _conv_ws(d, x.device)
    timed = KERNEL_TIMER is not None and d.C_in > 1 and d.C_out > 1
    t0 = KERNEL_TIMER.begin() if timed else None
    _lib.tag(_conv_label(d, "forward"), _gemm_flops(d))
    _lib.call("nsg_conv_forward", byref(d), _p(x), _p(w_fwd), _p(bias), _p(y), c_int32(flags), _p(ws), c_size_t(nb), _stream())
    if timed:
        KERNEL_TIMER.end("gather_gemm_f32", t0, _gemm_flops(d))
    return y


def conv_forward_bnstats(d: ConvDesc, x, w_fwd, bias, flags=0, running_mean=None, running_var=None, eps=BN_EPS,
                         momentum=BN_MOMENTUM, out=None):
    """conv forward + training-mode BatchNorm statistics of the output in one pass -> (y, mean, invstd)."""
    _chk(x, "x", _in_dtype(d))
    if tuple(x.shape) != (d.B, d.IH, d.IW, d.C_in):
        raise _lib.NsgError(f"conv_forward_bnstats: input shape {tuple(x.shape)} does not match descriptor {d.key()}")
    y = out if out is not None else torch.empty(d.B, d.OH, d.OW, d.C_out, dtype=_out_dtype(d, flags), device=x.device)
    mean = torch.empty(d.C_out, dtype=torch.float32, device=x.device)
    invstd = torch.empty(d.C_out, dtype=torch.float32, device=x.device)
    ws, nb = _conv_ws(d, x.device)
    timed = KERNEL_TIMER is not None and d.C_in > 1 and d.C_out > 1
    t0 = KERNEL_TIMER.begin() if timed else None
    _lib.tag(_conv_label(d, "forward"), _gemm_flops(d))
    _lib.call("nsg_conv_forward_bnstats", byref(d), _p(x), _p(w_fwd), _p(bias), _p(y), c_int32(flags), c_float(eps),
              c_float(momentum), _p(mean), _p(invstd), _p(running_mean), _p(running_var), _p(ws), c_size_t(nb), _stream())
    if timed:
        KERNEL_TIMER.end("gather_gemm_f32", t0, _gemm_flops(d))
    return y, mean, invstd


def conv_dgrad(d: ConvDesc, dy, w_dgrad, out=None, add=None, relu_x=None):
    """dx = conv_dgrad(dy); with add / relu_x: dx = (conv_dgrad(dy) + add) * (relu_x > 0) in the same kernel."""
    _chk(dy, "dy", _out_dtype(d))
    if tuple(dy.shape) != (d.B, d.OH, d.OW, d.C_out):
        raise _lib.NsgError(f"conv_dgrad: dy shape {tuple(dy.shape)} does not match descriptor {d.key()}")
    dx = out if out is not None else torch.empty(d.B, d.IH, d.IW, d.C_in, dtype=_in_dtype(d), device=dy.device)
    ws, nb = _conv_ws(d, dy.device)
    timed = KERNEL_TIMER is not None and d.C_in > 1 and d.C_out > 1
    t0 = KERNEL_TIMER.begin() if timed else None
    if add is not None or relu_x is not None:
        for t, nm in ((add, "add"), (relu_x, "relu_x")):
            if t is not None:
                _chk(t, nm, dx.dtype)
                if t.shape != dx.shape:
                    raise _lib.NsgError(f"conv_dgrad: {nm} shape {tuple(t.shape)} does not match dx {tuple(dx.shape)}")
        _lib.tag(_conv_label(d, "dgrad"), _gemm_flops(d))
        _lib.call("nsg_conv_dgrad_relu_add", byref(d), _p(dy), _p(w_dgrad), _p(add), _p(relu_x), _p(dx), c_int32(0), _p(ws),
                  c_size_t(nb), _stream())
    else:
        _lib.tag(_conv_label(d, "dgrad"), _gemm_flops(d))
        _lib.call("nsg_conv_dgrad", byref(d), _p(dy), _p(w_dgrad), _p(dx), c_int32(0), _p(ws), c_size_t(nb), _stream())
    if timed:
        KERNEL_TIMER.end("gather_gemm_f32", t0, _gemm_flops(d))
    return dx


def conv_wgrad(d: ConvDesc, x, dy, w_shape, flags=0, dw=None, dbias=None, want_bias=True):
    _chk(x, "x", _in_dtype(d)); _chk(dy, "dy", _out_dtype(d))
    if dw is None:
        dw = torch.empty(w_shape, dtype=torch.float32, device=x.device)
    if dbias is None and want_bias:
        dbias = torch.empty(d.C_out, dtype=torch.float32, device=x.device)
    ws, nb = _conv_ws(d, x.device)
    _lib.tag(_conv_label(d, "wgrad"), _gemm_flops(d))
    _lib.call("nsg_conv_wgrad", byref(d), _p(x), _p(dy), _p(dw), _p(dbias), c_int32(flags), _p(ws), c_size_t(nb), _stream())
    return dw, dbias


# ------------------------------------------------------------------------------------------------
# batch norm over [M][C]
# ------------------------------------------------------------------------------------------------
def bn_stats(x, C, running_mean=None, running_var=None, eps=BN_EPS, momentum=BN_MOMENTUM):
    _chk(x, "x", None)
    M = x.numel() // C
    mean = torch.empty(C, dtype=torch.float32, device=x.device)
    invstd = torch.empty(C, dtype=torch.float32, device=x.device)
    nb = _lib.query("nsg_bn_workspace_bytes", c_int64(M), c_int32(C))
    ws = WS.get(nb, x.device)
    _lib.tag("bn_stats", 0, x.numel() * _es(x))
    _lib.call("nsg_bn_stats", _p(x), c_int64(M), c_int32(C), c_int32(nsg_dtype(x.dtype)), c_float(eps), c_float(momentum), _p(mean), _p(invstd),
              _p(running_mean), _p(running_var), _p(ws), c_size_t(nb), _stream())
    return mean, invstd


def bn_eval_stats(running_mean, running_var, eps=BN_EPS):
    C = running_mean.numel()
    mean = torch.empty(C, dtype=torch.float32, device=running_mean.device)
    invstd = torch.empty(C, dtype=torch.float32, device=running_mean.device)
    _lib.call("nsg_bn_eval_stats", _p(running_mean), _p(running_var), c_int32(C), c_float(eps), _p(mean), _p(invstd), _stream())
    return mean, invstd


def bn_apply(x, mean, invstd, gamma, beta, relu=False, residual=None, relu_residual=False, out=None, out_dtype=None,
             relu_out=False):
    """out_dtype: storage type of y (default: x's); the residual must have x's type.  relu_out: max(0,.) of the
    final value (the consumer's leading ReLU applied at the producer)."""
    _chk(x, "x", None)
    if residual is not None:
        _chk(residual, "residual", x.dtype)
    C = mean.numel()
    M = x.numel() // C
    y = out if out is not None else torch.empty(x.shape, dtype=out_dtype or x.dtype, device=x.device)
    _lib.tag("bn_apply", 0, x.numel() * _es(x) * (2 if residual is not None else 1) + y.numel() * _es(y))
    _lib.call("nsg_bn_apply", _p(x), _p(mean), _p(invstd), _p(gamma), _p(beta), _p(residual), _p(y), c_int64(M), c_int32(C),
              c_int32((1 if relu else 0) | (2 if relu_out else 0)), c_int32(1 if relu_residual else 0), c_int32(nsg_dtype(x.dtype)),
              c_int32(nsg_dtype(y.dtype)),
              _stream())
    return y


def bn_backward(x, y_relu, dy, mean, invstd, gamma, dgamma=None, dbeta=None, out=None, dx_colsum=None, relu_beta=None):
    """dx_colsum: optional [C] tensor receiving the column sums of dx (= bias gradient of the conv in front).
    ReLU mask of a fused BatchNorm+ReLU forward: relu_beta (the forward's beta: mask re-derived from x) or
    y_relu (the stored forward output)."""
    _chk(x, "x", None); _chk(dy, "dy", x.dtype)
    if y_relu is not None:
        _chk(y_relu, "y_relu", x.dtype)
    C = mean.numel()
    M = x.numel() // C
    dx = out if out is not None else torch.empty_like(x)
    if dgamma is None:
        dgamma = torch.empty(C, dtype=torch.float32, device=x.device)
    if dbeta is None:
        dbeta = torch.empty(C, dtype=torch.float32, device=x.device)
    nb = _lib.query("nsg_bn_workspace_bytes", c_int64(M), c_int32(C))
    ws = WS.get(nb, x.device)
    _lib.tag("bn_backward (sums pass + apply pass)", 0, 5.0 * x.numel() * _es(x))
    _lib.call("nsg_bn_backward", _p(x), _p(y_relu), _p(dy), _p(mean), _p(invstd), _p(gamma), _p(relu_beta), _p(dx), _p(dgamma), _p(dbeta),
              _p(dx_colsum), c_int64(M), c_int32(C), c_int32(nsg_dtype(x.dtype)), _p(ws), c_size_t(nb), _stream())
    return dx, dgamma, dbeta


# ------------------------------------------------------------------------------------------------
# encoder.0-2 as one operator: Conv2d(1, C, 4, 2, 1) -> BatchNorm2d -> ReLU   (src/models.py:165-167)
# ------------------------------------------------------------------------------------------------
C1_MOMENTS = 273      # include/nsg.h: NSG_C1_MOMENTS


def c1conv_bn_relu_forward(img, w, bias, gamma, beta, running_mean=None, running_var=None, training=True, eps=1e-5, momentum=0.1,
                           mean=None, invstd=None, out_dtype=torch.float32, moments=None):
    """img fp32 (B, H, W) or (B, H, W, 1); w the Conv2d parameter (C, 1, 4, 4) fp32.  Returns (y NHWC (B, H/2, W/2, C) of
    out_dtype, mean, invstd).  training=False: mean / invstd must be given (bn_eval_stats).  The conv output itself is
    never stored (nsg.h: nsg_c1conv_bn_relu_forward).  moments: optional float64 tensor of C1_MOMENTS elements that receives the
    image's tap moments in training mode (hand it to c1conv_bn_relu_backward: it then skips recomputing them)."""
    _chk(img, "img", torch.float32); _chk(w, "w", torch.float32)
    if moments is not None and (moments.dtype != torch.float64 or moments.numel() != C1_MOMENTS or not moments.is_contiguous()):
        raise ValueError("c1conv_bn_relu_forward: moments must be a contiguous float64 tensor of C1_MOMENTS elements")
    B, H, W = img.shape[0], img.shape[1], img.shape[2]
    C = w.shape[0]
    if w.numel() != C * 16 or img.numel() != B * H * W:
        raise ValueError("c1conv_bn_relu_forward: img must be single-channel and w (C, 1, 4, 4)")
    if training:
        mean = torch.empty(C, dtype=torch.float32, device=img.device)
        invstd = torch.empty(C, dtype=torch.float32, device=img.device)
    elif mean is None or invstd is None:
        raise ValueError("c1conv_bn_relu_forward: eval mode needs mean and invstd")
    y = torch.empty((B, H // 2, W // 2, C), dtype=out_dtype, device=img.device)
    nb = _lib.query("nsg_c1conv_bn_workspace_bytes", c_int32(C))
    ws = WS.get(nb, img.device)
    _lib.tag("c1conv_bn_relu_forward (fused input layer)", 2.0 * 16 * y.numel() * (2 if training else 1), 4.0 * img.numel() * (2 if training else 1) + y.numel() * _es(y))
    _lib.call("nsg_c1conv_bn_relu_forward", _p(img), _p(w), _p(bias), _p(gamma), _p(beta), _p(mean), _p(invstd), _p(running_mean),
              _p(running_var), c_float(eps), c_float(momentum), c_int32(1 if training else 0), _p(y), c_int32(nsg_dtype(out_dtype)),
              c_int32(B), c_int32(H), c_int32(W), c_int32(C), _p(ws), c_size_t(nb), _p(moments), _stream())
    return y, mean, invstd


def c1conv_bn_relu_backward(img, w, bias, gamma, beta, mean, invstd, dy, dw=None, dbias=None, dgamma=None, dbeta=None, moments=None):
    """Parameter gradients (dw (C, 1, 4, 4), dbias, dgamma, dbeta) of the fused layer from dy (B, H/2, W/2, C).
    moments: what the training forward wrote (same image), or None (recomputed)."""
    _chk(img, "img", torch.float32); _chk(w, "w", torch.float32); _chk(dy, "dy", None)
    B, H, W = img.shape[0], img.shape[1], img.shape[2]
    C = w.shape[0]
    if dy.numel() != B * (H // 2) * (W // 2) * C:
        raise ValueError("c1conv_bn_relu_backward: dy does not match (B, H/2, W/2, C)")
    dev = img.device
    dw = dw if dw is not None else torch.empty_like(w)
    dbias = dbias if dbias is not None else torch.empty(C, dtype=torch.float32, device=dev)
    dgamma = dgamma if dgamma is not None else torch.empty(C, dtype=torch.float32, device=dev)
    dbeta = dbeta if dbeta is not None else torch.empty(C, dtype=torch.float32, device=dev)
    nb = _lib.query("nsg_c1conv_bn_workspace_bytes", c_int32(C))
    ws = WS.get(nb, dev)
    passes = 1.0 if dy.dtype == torch.bfloat16 else 2.0          # bf16: one pass over dy (tap moments), fp32: sums pass + gradient pass
    _lib.tag("c1conv_bn_relu_backward (fused input layer)", 2.0 * 16 * dy.numel() * 3, passes * (4.0 * img.numel() + dy.numel() * _es(dy)))
    _lib.call("nsg_c1conv_bn_relu_backward", _p(img), _p(w), _p(bias), _p(gamma), _p(beta), _p(mean), _p(invstd), _p(dy),
              c_int32(nsg_dtype(dy.dtype)), _p(dw), _p(dbias), _p(dgamma), _p(dbeta), c_int32(B), c_int32(H), c_int32(W), c_int32(C),
              _p(ws), c_size_t(nb), _p(moments), _stream())
    return dw, dbias, dgamma, dbeta


# ------------------------------------------------------------------------------------------------
# the ResBlock's 1x1 conv with the BatchNorm work around it folded in   (src/models.py:151-155)
# ------------------------------------------------------------------------------------------------
def bn_relu_conv1x1_supported(dtype, C) -> bool:
    return bool(_lib.query("nsg_bn_relu_conv1x1_supported", c_int32(nsg_dtype(dtype)), c_int32(C)))


def _ws_1x1(M, C, dev):
    nb = _lib.query("nsg_bn_relu_conv1x1_workspace_bytes", c_int64(M), c_int32(C))
    return WS.get(nb, dev), nb


def bn_relu_conv1x1_forward(x, mean, invstd, gamma, beta, w, bias):
    """y = relu(bn(x)) * w^T + bias on NHWC rows; relu(bn(x)) is never stored.  w (C, C, 1, 1) fp32."""
    _chk(x, "x", None); _chk(w, "w", torch.float32)
    C = x.shape[-1]
    M = x.numel() // C
    y = torch.empty_like(x)
    ws, nb = _ws_1x1(M, C, x.device)
    _lib.tag("flat_gemm 1x1 forward (bn+relu on load)", 2.0 * M * C * C, 2.0 * x.numel() * _es(x))
    _lib.call("nsg_bn_relu_conv1x1_forward", _p(x), _p(mean), _p(invstd), _p(gamma), _p(beta), _p(w), _p(bias), _p(y), c_int64(M), c_int32(C),
              c_int32(nsg_dtype(x.dtype)), _p(ws), c_size_t(nb), _stream())
    return y


def bn_relu_conv1x1_forward_bnstats(x, mean, invstd, gamma, beta, w, bias, running_mean=None, running_var=None, eps=BN_EPS,
                                    momentum=BN_MOMENTUM):
    """bn_relu_conv1x1_forward plus the batch statistics of its output (for the BatchNorm that follows), taken from the kernel's
    store phase: returns (y, mean_y, invstd_y); running statistics updated in place."""
    _chk(x, "x", None); _chk(w, "w", torch.float32)
    C = x.shape[-1]
    M = x.numel() // C
    y = torch.empty_like(x)
    mean_y = torch.empty(C, dtype=torch.float32, device=x.device)
    invstd_y = torch.empty(C, dtype=torch.float32, device=x.device)
    ws, nb = _ws_1x1(M, C, x.device)
    _lib.tag("flat_gemm 1x1 forward (bn+relu on load, stats out)", 2.0 * M * C * C, 2.0 * x.numel() * _es(x))
    _lib.call("nsg_bn_relu_conv1x1_forward_bnstats", _p(x), _p(mean), _p(invstd), _p(gamma), _p(beta), _p(w), _p(bias), _p(y), c_float(eps),
              c_float(momentum), _p(mean_y), _p(invstd_y), _p(running_mean), _p(running_var), c_int64(M), c_int32(C),
              c_int32(nsg_dtype(x.dtype)), _p(ws), c_size_t(nb), _stream())
    return y, mean_y, invstd_y


def bn_relu_conv1x1_wgrad(x, mean, invstd, gamma, beta, dy, dw=None):
    """dw (C, C, 1, 1) = dy^T relu(bn(x)) with the activation rebuilt from x on the operand's way into the MFMA."""
    _chk(x, "x", None); _chk(dy, "dy", x.dtype)
    C = x.shape[-1]
    M = x.numel() // C
    dw = dw if dw is not None else torch.empty((C, C, 1, 1), dtype=torch.float32, device=x.device)
    ws, nb = _ws_1x1(M, C, x.device)
    _lib.tag("wgrad_gemm 1x1 (bn+relu on load)", 2.0 * M * C * C, 2.0 * x.numel() * _es(x))
    _lib.call("nsg_bn_relu_conv1x1_wgrad", _p(x), _p(mean), _p(invstd), _p(gamma), _p(beta), _p(dy), _p(dw), c_int64(M), c_int32(C),
              c_int32(nsg_dtype(x.dtype)), _p(ws), c_size_t(nb), _stream())
    return dw


def bn_backward_sums(x, dy, mean, invstd, gamma, dgamma=None, dbeta=None, relu_beta=None):
    """The reduction half of bn_backward: (dgamma, dbeta)."""
    _chk(x, "x", None); _chk(dy, "dy", x.dtype)
    C = mean.numel()
    M = x.numel() // C
    dgamma = dgamma if dgamma is not None else torch.empty(C, dtype=torch.float32, device=x.device)
    dbeta = dbeta if dbeta is not None else torch.empty(C, dtype=torch.float32, device=x.device)
    nb = _lib.query("nsg_bn_workspace_bytes", c_int64(M), c_int32(C))
    ws = WS.get(nb, x.device)
    _lib.tag("bn_backward_sums", 0, 2.0 * x.numel() * _es(x))
    _lib.call("nsg_bn_backward_sums", _p(x), _p(None), _p(dy), _p(mean), _p(invstd), _p(gamma), _p(relu_beta), _p(dgamma), _p(dbeta),
              c_int64(M), c_int32(C), c_int32(nsg_dtype(x.dtype)), _p(ws), c_size_t(nb), _stream())
    return dgamma, dbeta


def bn_backward_conv1x1_dgrad(h, dy, mean, invstd, gamma, dgamma, dbeta, w, dh_colsum=None, prev=None, prev_dgamma=None, prev_dbeta=None):
    """(dh, dx): dh = the BatchNorm's input gradient (no ReLU) at input h, dx = dh * w (the data gradient of the 1x1 conv that
    wrote h); dh_colsum: optional [C] tensor receiving dh's column sums (that conv's bias gradient).
    prev = (prev_x, mean, invstd, gamma, beta) of the BatchNorm + ReLU in front of the conv: its backward sums over (prev_x, dx)
    are formed while dx is written; returns (dh, dx, prev_dgamma, prev_dbeta) then."""
    _chk(h, "h", None); _chk(dy, "dy", h.dtype); _chk(w, "w", torch.float32)
    C = h.shape[-1]
    M = h.numel() // C
    dh, dx = torch.empty_like(h), torch.empty_like(h)
    ws, nb = _ws_1x1(M, C, h.device)
    px = pm = pi = pg = pb = None
    if prev is not None:
        px, pm, pi, pg, pb = prev
        _chk(px, "prev_x", h.dtype)
        prev_dgamma = prev_dgamma if prev_dgamma is not None else torch.empty(C, dtype=torch.float32, device=h.device)
        prev_dbeta = prev_dbeta if prev_dbeta is not None else torch.empty(C, dtype=torch.float32, device=h.device)
    _lib.tag("flat_gemm 1x1 dgrad (bn backward on load)", 2.0 * M * C * C, (5.0 if prev is not None else 4.0) * h.numel() * _es(h))
    _lib.call("nsg_bn_backward_conv1x1_dgrad", _p(h), _p(dy), _p(mean), _p(invstd), _p(gamma), _p(dgamma), _p(dbeta), _p(w), _p(dh), _p(dx),
              _p(dh_colsum), _p(px), _p(pm), _p(pi), _p(pg), _p(pb), _p(prev_dgamma if prev is not None else None),
              _p(prev_dbeta if prev is not None else None), c_int64(M), c_int32(C), c_int32(nsg_dtype(h.dtype)), _p(ws), c_size_t(nb), _stream())
    if prev is not None:
        return dh, dx, prev_dgamma, prev_dbeta
    return dh, dx


def bn_backward_conv1x1_dgrad_wgrad_supported(dtype, C) -> bool:
    return bool(_lib.query("nsg_bn_backward_conv1x1_dgrad_wgrad_supported", c_int32(nsg_dtype(dtype)), c_int32(C)))


def bn_backward_conv1x1_dgrad_wgrad(h, dy, mean, invstd, gamma, dgamma, dbeta, w, prev, dh_colsum=None, dw=None, prev_dgamma=None,
                                    prev_dbeta=None):
    """bn_backward_conv1x1_dgrad(..., prev=...) and bn_relu_conv1x1_wgrad in one pass over the tensors: returns
    (dx, dw, prev_dgamma, prev_dbeta); dh is not stored (include/nsg.h: nsg_bn_backward_conv1x1_dgrad_wgrad)."""
    _chk(h, "h", None); _chk(dy, "dy", h.dtype); _chk(w, "w", torch.float32)
    C = h.shape[-1]
    M = h.numel() // C
    px, pm, pi, pg, pb = prev
    _chk(px, "prev_x", h.dtype)
    dev = h.device
    dx = torch.empty_like(h)
    dw = dw if dw is not None else torch.empty_like(w)
    prev_dgamma = prev_dgamma if prev_dgamma is not None else torch.empty(C, dtype=torch.float32, device=dev)
    prev_dbeta = prev_dbeta if prev_dbeta is not None else torch.empty(C, dtype=torch.float32, device=dev)
    nb = _lib.query("nsg_bn_backward_conv1x1_dgrad_wgrad_workspace_bytes", c_int64(M), c_int32(C))
    ws = WS.get(nb, dev)
    _lib.tag("flat_gemm 1x1 dgrad + wgrad (bn backward on load)", 4.0 * M * C * C, 4.0 * h.numel() * _es(h))
    _lib.call("nsg_bn_backward_conv1x1_dgrad_wgrad", _p(h), _p(dy), _p(mean), _p(invstd), _p(gamma), _p(dgamma), _p(dbeta), _p(w), _p(dx), _p(dw),
              _p(dh_colsum), _p(px), _p(pm), _p(pi), _p(pg), _p(pb), _p(prev_dgamma), _p(prev_dbeta), c_int64(M), c_int32(C),
              c_int32(nsg_dtype(h.dtype)), _p(ws), c_size_t(nb), _stream())
    return dx, dw, prev_dgamma, prev_dbeta


def bn_backward_apply(x, dy, mean, invstd, gamma, dgamma, dbeta, relu_beta=None, dx_colsum=None):
    """The apply half of bn_backward with dgamma / dbeta given: dx."""
    _chk(x, "x", None); _chk(dy, "dy", x.dtype)
    C = mean.numel()
    M = x.numel() // C
    dx = torch.empty_like(x)
    nb = _lib.query("nsg_bn_workspace_bytes", c_int64(M), c_int32(C))
    ws = WS.get(nb, x.device)
    _lib.tag("bn_backward_apply", 0, 3.0 * x.numel() * _es(x))
    _lib.call("nsg_bn_backward_apply", _p(x), _p(None), _p(dy), _p(mean), _p(invstd), _p(gamma), _p(relu_beta), _p(dgamma), _p(dbeta), _p(dx),
              _p(dx_colsum), c_int64(M), c_int32(C), c_int32(nsg_dtype(x.dtype)), _p(ws), c_size_t(nb), _stream())
    return dx


# ------------------------------------------------------------------------------------------------
# decoder.4-7 as one operator: BatchNorm2d -> ReLU -> ConvTranspose2d(C, 1, 4, 2, 1) [-> Tanh]   (src/models.py:180-183)
# ------------------------------------------------------------------------------------------------
def bn_relu_c1convt_supported(dtype, C) -> bool:
    return bool(_lib.query("nsg_bn_relu_c1convt_supported", c_int32(nsg_dtype(dtype)), c_int32(C)))


def bn_relu_c1convt_forward(u, mean, invstd, gamma, beta, w, bias, tanh=True):
    """u NHWC (B, H, W, C) = the BatchNorm input; w the ConvTranspose2d parameter (C, 1, 4, 4) fp32.  Returns the fp32 image
    (B, 2H, 2W, 1).  relu(bn(u)) is never stored (nsg.h: nsg_bn_relu_c1convt_forward)."""
    _chk(u, "u", None); _chk(w, "w", torch.float32)
    B, H, W, C = u.shape
    y = torch.empty((B, 2 * H, 2 * W, 1), dtype=torch.float32, device=u.device)
    nb = _lib.query("nsg_bn_relu_c1convt_workspace_bytes", c_int32(B), c_int32(H), c_int32(W), c_int32(C))
    ws = WS.get(nb, u.device)
    _lib.tag("bn_relu_c1convt_forward (fused output layer)", 2.0 * 16 * u.numel(), u.numel() * _es(u) + 4.0 * y.numel())
    _lib.call("nsg_bn_relu_c1convt_forward", _p(u), c_int32(nsg_dtype(u.dtype)), _p(mean), _p(invstd), _p(gamma), _p(beta), _p(w), _p(bias),
              _p(y), c_int32(NSG_TANH_OUT if tanh else 0), c_int32(B), c_int32(H), c_int32(W), c_int32(C), _p(ws), c_size_t(nb), _stream())
    return y


def bn_relu_c1convt_forward_mse(u, mean, invstd, gamma, beta, w, bias, target, grad_scale=1.0, want_image=False, dbias=None):
    """bn_relu_c1convt_forward(tanh=True) + mse_padded + tanh_backward in one pass over the tap products: target fp32
    (B, 2H, T[, 1]) with T >= 2W.  Returns (loss (1,), dpre (B, 2H, 2W, 1) = the gradient w.r.t. the Tanh's input, x_tilde or None).
    dbias: optional (1,) tensor receiving sum(dpre), the transposed conv's bias gradient."""
    _chk(u, "u", None); _chk(w, "w", torch.float32); _chk(target, "target", torch.float32)
    B, H, W, C = u.shape
    T = target.numel() // (B * 2 * H)
    if target.numel() != B * 2 * H * T or T < 2 * W:
        raise ValueError("bn_relu_c1convt_forward_mse: target must be (B, 2H, T) with T >= 2W")
    dev = u.device
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    dpre = torch.empty((B, 2 * H, 2 * W, 1), dtype=torch.float32, device=dev)
    y = torch.empty((B, 2 * H, 2 * W, 1), dtype=torch.float32, device=dev) if want_image else None
    nb = _lib.query("nsg_bn_relu_c1convt_workspace_bytes", c_int32(B), c_int32(H), c_int32(W), c_int32(C))
    ws = WS.get(nb, dev)
    _lib.tag("bn_relu_c1convt_forward_mse (fused output layer + loss)", 2.0 * 16 * u.numel(),
             u.numel() * _es(u) + 4.0 * (2 * dpre.numel() + (dpre.numel() if want_image else 0)))
    _lib.call("nsg_bn_relu_c1convt_forward_mse", _p(u), c_int32(nsg_dtype(u.dtype)), _p(mean), _p(invstd), _p(gamma), _p(beta), _p(w), _p(bias),
              _p(y), _p(target), c_int32(T), c_float(grad_scale), _p(loss), _p(dpre), _p(dbias), c_int32(B), c_int32(H), c_int32(W), c_int32(C),
              _p(ws), c_size_t(nb), _stream())
    return loss, dpre, y


def bn_relu_c1convt_backward(u, mean, invstd, gamma, beta, w, dy, dw=None, dbias=None, dgamma=None, dbeta=None, du_colsum=None,
                             want_dbias=True):
    """dy: fp32 gradient image (B, 2H, 2W[, 1]) w.r.t. the transposed conv's output (before the tanh).
    Returns (du like u, dw (C, 1, 4, 4), dbias (1,), dgamma, dbeta); du_colsum: optional [C] tensor receiving the column
    sums of du (the bias gradient of the conv in front of the BatchNorm).  want_dbias=False: the bias gradient (sum of dy) is
    not formed (bn_relu_c1convt_forward_mse already gave it); dbias is returned as passed."""
    _chk(u, "u", None); _chk(w, "w", torch.float32); _chk(dy, "dy", torch.float32)
    B, H, W, C = u.shape
    if dy.numel() != B * 4 * H * W:
        raise ValueError("bn_relu_c1convt_backward: dy does not match (B, 2H, 2W)")
    dev = u.device
    du = torch.empty_like(u)
    dw = dw if dw is not None else torch.empty_like(w)
    if want_dbias:
        dbias = dbias if dbias is not None else torch.empty(1, dtype=torch.float32, device=dev)
    dgamma = dgamma if dgamma is not None else torch.empty(C, dtype=torch.float32, device=dev)
    dbeta = dbeta if dbeta is not None else torch.empty(C, dtype=torch.float32, device=dev)
    nb = _lib.query("nsg_bn_relu_c1convt_workspace_bytes", c_int32(B), c_int32(H), c_int32(W), c_int32(C))
    ws = WS.get(nb, dev)
    _lib.tag("bn_relu_c1convt_backward (fused output layer)", 2.0 * 16 * u.numel() * 3, 3.0 * u.numel() * _es(u) + 2.0 * 4.0 * dy.numel())
    _lib.call("nsg_bn_relu_c1convt_backward", _p(u), c_int32(nsg_dtype(u.dtype)), _p(mean), _p(invstd), _p(gamma), _p(beta), _p(w), _p(dy),
              _p(du), _p(du_colsum), _p(dw), _p(dbias if want_dbias else None), _p(dgamma), _p(dbeta), c_int32(B), c_int32(H), c_int32(W), c_int32(C), _p(ws),
              c_size_t(nb), _stream())
    return du, dw, dbias, dgamma, dbeta


# ------------------------------------------------------------------------------------------------
# element-wise / losses / optimiser
# ------------------------------------------------------------------------------------------------
def relu_backward_add(a, b, x, out=None):
    _chk(a, "a", x.dtype)
    if b is not None:
        _chk(b, "b", x.dtype)
    dx = out if out is not None else torch.empty_like(x)
    _lib.call("nsg_relu_backward_add", _p(a), _p(b), _p(x), _p(dx), c_int64(x.numel()), c_int32(nsg_dtype(x.dtype)), _stream())
    return dx


def convert(src, dtype, out=None, relu=False):
    """Change of storage type (fp32 <-> bf16), optionally with max(0,.), by the library's own kernel."""
    _chk(src, "src", None)
    if src.dtype == dtype and out is None and not relu:
        return src
    dst = out if out is not None else torch.empty(src.shape, dtype=dtype, device=src.device)
    _lib.tag("convert", 0, src.numel() * _es(src) + dst.numel() * _es(dst))
    _lib.call("nsg_convert", _p(src), c_int32(nsg_dtype(src.dtype)), _p(dst), c_int32(nsg_dtype(dst.dtype)), c_int64(src.numel()),
              c_int32(1 if relu else 0), _stream())
    return dst


def tanh_backward(g, y, out=None):
    dx = out if out is not None else torch.empty_like(y)
    _lib.tag("tanh_backward", 0, 12.0 * y.numel())
    _lib.call("nsg_tanh_backward", _p(g), _p(y), _p(dx), c_int64(y.numel()), _stream())
    return dx


def add(a, b, out=None):
    y = out if out is not None else torch.empty_like(a)
    _lib.call("nsg_add", _p(a), _p(b), _p(y), c_int64(a.numel()), _stream())
    return y


def add_per_clip(x, rows, out=None, out_dtype=torch.float32):
    """x fp32 NHWC (B,H,W,C) + rows (B,C) broadcast over each clip's pixels -> y of out_dtype."""
    _chk(x, "x"); _chk(rows, "rows")
    B, C = rows.shape
    y = out if out is not None else torch.empty(x.shape, dtype=out_dtype, device=x.device)
    _lib.call("nsg_add_per_clip", _p(x), _p(rows), _p(y), c_int32(B), c_int64(x.numel() // (B * C)), c_int32(C),
              c_int32(nsg_dtype(y.dtype)), _stream())
    return y


def clip_colsum(x, B):
    """x NHWC (B,H,W,C) -> (B,C): per-clip sum over pixels."""
    _chk(x, "x", None)
    C = x.shape[-1]
    out = torch.empty(B, C, dtype=torch.float32, device=x.device)
    nb = _lib.query("nsg_clip_colsum_workspace_bytes", c_int32(B), c_int32(C))
    ws = WS.get(nb, x.device)
    _lib.call("nsg_clip_colsum", _p(x), c_int32(nsg_dtype(x.dtype)), c_int32(B), c_int64(x.numel() // (B * C)), c_int32(C), _p(out), _p(ws),
              c_size_t(nb), _stream())
    return out


def mse_padded(a, c, rows, wa, wc, grad_scale=1.0, want_grad=True):
    """mean((pad(a) - c)^2) with a zero-padded from width wa to wc (train.py:118-129)."""
    loss = torch.empty(1, dtype=torch.float32, device=a.device)
    da = torch.empty_like(a) if want_grad else None
    nb = _lib.query("nsg_reduce_workspace_bytes", c_int64(rows * wc))
    ws = WS.get(nb, a.device)
    _lib.tag("mse_padded (loss + gradient)", 0, 4.0 * (a.numel() * (2 if want_grad else 1) + c.numel()))
    _lib.call("nsg_mse_padded", _p(a), _p(c), c_int64(rows), c_int32(wa), c_int32(wc), c_float(grad_scale), _p(loss), _p(da),
              _p(ws), c_size_t(nb), _stream())
    return loss, da


def vq_losses(z, q, dz_scale=1.0, dq_scale=1.0, dz_add=None, want_dz=True, want_dq=True, grad_dtype=torch.float32):
    """mean((q - z)^2) and its two one-sided gradients (train.py:131,133).  z, q, dq fp32; dz (and dz_add,
    the straight-through gradient coming out of the decoder) in grad_dtype."""
    _chk(z, "z"); _chk(q, "q")
    if dz_add is not None:
        _chk(dz_add, "dz_add", grad_dtype)
    n = z.numel()
    loss = torch.empty(1, dtype=torch.float32, device=z.device)
    dz = torch.empty(z.shape, dtype=grad_dtype, device=z.device) if want_dz else None
    dq = torch.empty_like(z) if want_dq else None
    nb = _lib.query("nsg_reduce_workspace_bytes", c_int64(n))
    ws = WS.get(nb, z.device)
    _lib.tag("vq_losses", 0, 4.0 * n * 2 + (dz.numel() * _es(dz) * (2 if dz_add is not None else 1) if dz is not None else 0) + (4.0 * n if dq is not None else 0))
    _lib.call("nsg_vq_losses", _p(z), _p(q), c_int64(n), c_float(dz_scale), c_float(dq_scale), _p(dz_add), _p(loss), _p(dz),
              _p(dq), c_int32(nsg_dtype(grad_dtype)), _p(ws), c_size_t(nb), _stream())
    return loss, dz, dq


def vq_losses_indexed_bn_supported(D) -> bool:
    return bool(_lib.query("nsg_vq_losses_indexed_bn_supported", c_int32(D)))


def vq_losses_indexed(z2d, codebook, idx, dz_scale=1.0, dz_add=None, want_dz=True, grad_dtype=torch.float32, bn=None, dgamma=None,
                      dbeta=None):
    """vq_losses with q = codebook[idx] read from the codebook itself: returns (loss, dz).  z2d (N, D) fp32.
    bn = (x, mean, invstd): dz is the incoming gradient of a BatchNorm with input x (N, D) of grad_dtype; returns
    (loss, dz, dgamma, dbeta) with that BatchNorm's backward sums (= bn_backward_sums(x, dz, ...)) formed while dz is written."""
    _chk(z2d, "z"); _chk(codebook, "codebook"); _chk(idx, "idx", torch.int64)
    if dz_add is not None:
        _chk(dz_add, "dz_add", grad_dtype)
    N, D = z2d.shape
    loss = torch.empty(1, dtype=torch.float32, device=z2d.device)
    dz = torch.empty(z2d.shape, dtype=grad_dtype, device=z2d.device) if want_dz else None
    if bn is not None:
        x, mean, invstd = bn
        _chk(x, "bn x", grad_dtype); _chk(mean, "bn mean"); _chk(invstd, "bn invstd")
        if dz is None or x.numel() != N * D:
            raise _lib.NsgError("vq_losses_indexed: bn= needs want_dz and a BatchNorm input of z's shape")
        dgamma = dgamma if dgamma is not None else torch.empty(D, dtype=torch.float32, device=z2d.device)
        dbeta = dbeta if dbeta is not None else torch.empty(D, dtype=torch.float32, device=z2d.device)
        nb = _lib.query("nsg_vq_losses_indexed_bn_workspace_bytes", c_int64(N), c_int32(D))
        ws = WS.get(nb, z2d.device)
        _lib.tag("vq_losses_indexed", 0, 4.0 * N * D + 8.0 * N + dz.numel() * _es(dz) * (3 if dz_add is not None else 2))
        _lib.call("nsg_vq_losses_indexed_bn", _p(z2d), _p(codebook), _p(idx), c_int64(N), c_int32(D), c_int32(codebook.shape[0]),
                  c_float(dz_scale), _p(dz_add), _p(loss), _p(dz), c_int32(nsg_dtype(grad_dtype)), _p(x), _p(mean), _p(invstd), _p(dgamma),
                  _p(dbeta), _p(ws), c_size_t(nb), _stream())
        return loss, dz, dgamma, dbeta
    nb = _lib.query("nsg_reduce_workspace_bytes", c_int64(N * D))
    ws = WS.get(nb, z2d.device)
    _lib.tag("vq_losses_indexed", 0, 4.0 * N * D + 8.0 * N + (dz.numel() * _es(dz) * (2 if dz_add is not None else 1) if dz is not None else 0))
    _lib.call("nsg_vq_losses_indexed", _p(z2d), _p(codebook), _p(idx), c_int64(N), c_int32(D), c_int32(codebook.shape[0]), c_float(dz_scale),
              _p(dz_add), _p(loss), _p(dz), c_int32(nsg_dtype(grad_dtype)), _p(ws), c_size_t(nb), _stream())
    return loss, dz


def adam_step(p, g, m, v, step, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0):
    _lib.tag("adam_step", 0, 28.0 * p.numel())
    _lib.call("nsg_adam_step", _p(p), _p(g), _p(m), _p(v), c_int64(p.numel()), c_float(lr), c_float(beta1), c_float(beta2),
              c_float(eps), c_int32(step), c_float(grad_scale), _stream())


# ------------------------------------------------------------------------------------------------
# latent prior (GatedPixelCNN): element-wise pieces
# ------------------------------------------------------------------------------------------------
def gated_activation(x, cond=None):
    """x (..., 2C) NHWC rows, cond (B, 2C) or None -> tanh(a) * sigmoid(b) of the channel halves, (..., C)."""
    _chk(x, "x")
    C2 = x.shape[-1]
    M = x.numel() // C2
    y = torch.empty(*x.shape[:-1], C2 // 2, dtype=torch.float32, device=x.device)
    rpc = 1
    if cond is not None:
        _chk(cond, "cond")
        if cond.shape[-1] != C2 or M % cond.shape[0] != 0:
            raise _lib.NsgError(f"gated_activation: cond {tuple(cond.shape)} does not match x {tuple(x.shape)}")
        rpc = M // cond.shape[0]
    _lib.call("nsg_gated_activation_forward", _p(x), _p(cond), _p(y), c_int64(M), c_int32(C2 // 2), c_int64(rpc), _stream())
    return y


def gated_activation_backward(x, cond, dy):
    _chk(x, "x"); _chk(dy, "dy")
    C2 = x.shape[-1]
    M = x.numel() // C2
    dx = torch.empty_like(x)
    rpc = M // cond.shape[0] if cond is not None else 1
    _lib.call("nsg_gated_activation_backward", _p(x), _p(cond), _p(dy), _p(dx), c_int64(M), c_int32(C2 // 2), c_int64(rpc), _stream())
    return dx


def cross_entropy(logits2d, target, want_grad=True, grad_scale=1.0):
    """mean cross-entropy of rows (M, K) against int64 targets (M,) -> (loss[1], dlogits or None)."""
    _chk(logits2d, "logits"); _chk(target, "target", torch.int64)
    M, K = logits2d.shape
    loss = torch.empty(1, dtype=torch.float32, device=logits2d.device)
    dl = torch.empty_like(logits2d) if want_grad else None
    nb = _lib.query("nsg_cross_entropy_workspace_bytes", c_int64(M))
    ws = WS.get(nb, logits2d.device)
    _lib.call("nsg_cross_entropy", _p(logits2d), _p(target), c_int64(M), c_int32(K), c_float(grad_scale), _p(loss), _p(dl), _p(ws),
              c_size_t(nb), _stream())
    return loss, dl
