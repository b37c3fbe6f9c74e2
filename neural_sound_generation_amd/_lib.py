"""ctypes binding of libnsg.so (the C ABI declared in include/nsg.h).

There is no fallback: if the HIP library is missing or a call fails, a RuntimeError is raised.
The oracle under oracle/ is never imported from here.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int32, c_int64, c_size_t, c_void_p

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libnsg.so")

NSG_RELU_IN = 1
NSG_TANH_OUT = 2
NSG_OUT_F32 = 8
NSG_RELU_OUT = 16
NSG_F32 = 0
NSG_BF16 = 1
NSG_VERSION = 102      # include/nsg.h NSG_VERSION this binding was written against (bumped on ANY signature change)


class ConvDesc(Structure):
    """struct nsg_conv_desc (include/nsg.h)."""
    _fields_ = [("B", c_int32), ("IH", c_int32), ("IW", c_int32), ("C_in", c_int32),
                ("OH", c_int32), ("OW", c_int32), ("C_out", c_int32),
                ("k", c_int32), ("stride", c_int32), ("pad", c_int32), ("transposed", c_int32), ("dtype", c_int32),
                ("k_w", c_int32), ("pad_w", c_int32)]

    def key(self):
        return tuple(getattr(self, f) for f, _ in self._fields_)


_P = c_void_p
_D = POINTER(ConvDesc)
# name -> (restype, argtypes); restype None means "int status, checked"
_SIGS = {
    "nsg_version": (c_int32, []),
    "nsg_last_error_string": (c_char_p, []),
    "nsg_vq_workspace_bytes": (c_size_t, [c_int64, c_int32, c_int32]),
    "nsg_vq_forward": (None, [_P, _P, c_int64, c_int32, c_int32, _P, _P, _P, _P, c_size_t, _P]),
    "nsg_vq_bf16x3_workspace_bytes": (c_size_t, [c_int64, c_int32, c_int32]),
    "nsg_vq_forward_bf16x3": (None, [_P, _P, c_int64, c_int32, c_int32, _P, _P, _P, _P, c_int32, _P, c_size_t, _P]),
    "nsg_vq_forward_bf16x3_cond": (None, [_P, _P, c_int64, c_int32, c_int32, _P, _P, _P, _P, c_int32, _P, c_int64, _P, c_size_t, _P]),
    "nsg_debug_vq_forward_valu": (None, [_P, _P, c_int64, c_int32, c_int32, _P, _P, _P, _P, c_size_t, _P]),
    "nsg_rowsumsq": (None, [_P, c_int64, c_int32, _P, _P]),
    "nsg_index_add_workspace_bytes": (c_size_t, [c_int64, c_int32, c_int32]),
    "nsg_index_add_rows": (None, [_P, _P, c_int64, c_int32, c_int32, _P, _P, _P, c_size_t, _P]),
    "nsg_index_add_rows_bf16x2": (None, [_P, _P, c_int64, c_int32, c_int32, _P, _P, _P, c_size_t, _P]),
    "nsg_index_add_sorted_workspace_bytes": (c_size_t, [c_int64, c_int32, c_int32]),
    "nsg_index_add_rows_sorted": (None, [_P, _P, c_int64, c_int32, c_int32, _P, _P, _P, c_size_t, _P]),
    "nsg_gather_rows": (None, [_P, _P, c_int64, c_int32, c_int32, _P, _P]),
    "nsg_vq_ema_update": (None, [_P, _P, _P, _P, _P, c_int32, c_int32, c_float, c_float, _P, _P]),
    "nsg_codebook_grad_from_sums": (None, [_P, _P, _P, c_int32, c_int32, c_float, _P, _P]),
    "nsg_increment_counters": (None, [_P, c_int32, _P]),
    "nsg_packed_weight_floats": (c_size_t, [_D]),
    "nsg_pack_conv_weights": (None, [_D, _P, _P, _P, _P]),
    "nsg_pack_conv_weights_batch": (None, [c_int32, _P, _P, _P, _P, _P]),
    "nsg_conv_workspace_bytes": (c_size_t, [_D]),
    "nsg_conv_forward": (None, [_D, _P, _P, _P, _P, c_int32, _P, c_size_t, _P]),
    "nsg_conv_forward_bnstats": (None, [_D, _P, _P, _P, _P, c_int32, c_float, c_float, _P, _P, _P, _P, _P, c_size_t, _P]),
    "nsg_conv_dgrad": (None, [_D, _P, _P, _P, c_int32, _P, c_size_t, _P]),
    "nsg_conv_dgrad_relu_add": (None, [_D, _P, _P, _P, _P, _P, c_int32, _P, c_size_t, _P]),
    "nsg_conv_wgrad": (None, [_D, _P, _P, _P, _P, c_int32, _P, c_size_t, _P]),
    "nsg_bn_workspace_bytes": (c_size_t, [c_int64, c_int32]),
    "nsg_bn_stats": (None, [_P, c_int64, c_int32, c_int32, c_float, c_float, _P, _P, _P, _P, _P, c_size_t, _P]),
    "nsg_bn_eval_stats": (None, [_P, _P, c_int32, c_float, _P, _P, _P]),
    "nsg_bn_apply": (None, [_P, _P, _P, _P, _P, _P, _P, c_int64, c_int32, c_int32, c_int32, c_int32, c_int32, _P]),
    "nsg_bn_backward": (None, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int64, c_int32, c_int32, _P, c_size_t, _P]),
    "nsg_c1conv_bn_workspace_bytes": (c_size_t, [c_int32]),
    "nsg_c1conv_bn_relu_forward": (None, [_P, _P, _P, _P, _P, _P, _P, _P, _P, c_float, c_float, c_int32, _P, c_int32, c_int32, c_int32, c_int32,
                                          c_int32, _P, c_size_t, _P, _P]),
    "nsg_c1conv_bn_relu_backward": (None, [_P, _P, _P, _P, _P, _P, _P, _P, c_int32, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, _P,
                                           c_size_t, _P, _P]),
    "nsg_bn_relu_c1convt_supported": (c_int32, [c_int32, c_int32]),
    "nsg_bn_relu_c1convt_workspace_bytes": (c_size_t, [c_int32, c_int32, c_int32, c_int32]),
    "nsg_bn_relu_c1convt_forward": (None, [_P, c_int32, _P, _P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, c_int32, _P, c_size_t, _P]),
    "nsg_bn_relu_c1convt_forward_mse": (None, [_P, c_int32, _P, _P, _P, _P, _P, _P, _P, _P, c_int32, c_float, _P, _P, _P, c_int32, c_int32, c_int32,
                                               c_int32, _P, c_size_t, _P]),
    "nsg_bn_relu_c1convt_backward": (None, [_P, c_int32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, _P,
                                            c_size_t, _P]),
    "nsg_bn_relu_conv1x1_supported": (c_int32, [c_int32, c_int32]),
    "nsg_bn_relu_conv1x1_workspace_bytes": (c_size_t, [c_int64, c_int32]),
    "nsg_bn_relu_conv1x1_forward": (None, [_P, _P, _P, _P, _P, _P, _P, _P, c_int64, c_int32, c_int32, _P, c_size_t, _P]),
    "nsg_bn_relu_conv1x1_forward_bnstats": (None, [_P, _P, _P, _P, _P, _P, _P, _P, c_float, c_float, _P, _P, _P, _P, c_int64, c_int32, c_int32, _P,
                                                   c_size_t, _P]),
    "nsg_bn_relu_conv1x1_wgrad": (None, [_P, _P, _P, _P, _P, _P, _P, c_int64, c_int32, c_int32, _P, c_size_t, _P]),
    "nsg_bn_backward_conv1x1_dgrad": (None, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int64, c_int32, c_int32, _P,
                                             c_size_t, _P]),
    "nsg_bn_backward_conv1x1_dgrad_wgrad_supported": (c_int32, [c_int32, c_int32]),
    "nsg_bn_backward_conv1x1_dgrad_wgrad_workspace_bytes": (c_size_t, [c_int64, c_int32]),
    "nsg_bn_backward_conv1x1_dgrad_wgrad": (None, [_P] * 18 + [c_int64, c_int32, c_int32, _P, c_size_t, _P]),
    "nsg_bn_backward_apply": (None, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int64, c_int32, c_int32, _P, c_size_t, _P]),
    "nsg_bn_backward_sums": (None, [_P, _P, _P, _P, _P, _P, _P, _P, _P, c_int64, c_int32, c_int32, _P, c_size_t, _P]),
    "nsg_relu_backward_add": (None, [_P, _P, _P, _P, c_int64, c_int32, _P]),
    "nsg_convert": (None, [_P, c_int32, _P, c_int32, c_int64, c_int32, _P]),
    "nsg_tanh_backward": (None, [_P, _P, _P, c_int64, _P]),
    "nsg_add": (None, [_P, _P, _P, c_int64, _P]),
    "nsg_add_per_clip": (None, [_P, _P, _P, c_int32, c_int64, c_int32, c_int32, _P]),
    "nsg_clip_colsum_workspace_bytes": (c_size_t, [c_int32, c_int32]),
    "nsg_clip_colsum": (None, [_P, c_int32, c_int32, c_int64, c_int32, _P, _P, c_size_t, _P]),
    "nsg_reduce_workspace_bytes": (c_size_t, [c_int64]),
    "nsg_mse_padded": (None, [_P, _P, c_int64, c_int32, c_int32, c_float, _P, _P, _P, c_size_t, _P]),
    "nsg_vq_losses_indexed": (None, [_P, _P, _P, c_int64, c_int32, c_int32, c_float, _P, _P, _P, c_int32, _P, c_size_t, _P]),
    "nsg_vq_losses_indexed_bn_supported": (c_int32, [c_int32]),
    "nsg_vq_losses_indexed_bn_workspace_bytes": (c_size_t, [c_int64, c_int32]),
    "nsg_vq_losses_indexed_bn": (None, [_P, _P, _P, c_int64, c_int32, c_int32, c_float, _P, _P, _P, c_int32, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "nsg_vq_losses": (None, [_P, _P, c_int64, c_float, c_float, _P, _P, _P, _P, c_int32, _P, c_size_t, _P]),
    "nsg_adam_step": (None, [_P, _P, _P, _P, c_int64, c_float, c_float, c_float, c_float, c_int32, c_float, _P]),
    "nsg_gated_activation_forward": (None, [_P, _P, _P, c_int64, c_int32, c_int64, _P]),
    "nsg_gated_activation_backward": (None, [_P, _P, _P, _P, c_int64, c_int32, c_int64, _P]),
    "nsg_cross_entropy_workspace_bytes": (c_size_t, [c_int64]),
    "nsg_cross_entropy": (None, [_P, _P, c_int64, c_int32, c_float, _P, _P, _P, c_size_t, _P]),
    "nsg_audio_mel_to_linear": (None, [_P, _P, _P, c_int32, c_int32, c_int32, c_int32, c_float, c_float, c_float, c_float, _P]),
    "nsg_audio_griffin_lim_workspace_bytes": (c_size_t, [c_int32, c_int32, c_int32]),
    "nsg_audio_griffin_lim": (None, [_P, _P, _P, c_int32, c_int32, c_int32, c_int32, c_int32, _P, c_size_t, _P]),
    "nsg_audio_stft": (None, [_P, _P, c_int32, c_int32, c_int32, c_int32, _P]),
    "nsg_audio_inv_preemphasis": (None, [_P, _P, c_int32, c_int32, c_float, _P]),
    "nsg_debug_dot": (None, [_P, _P, c_int32, c_int32, c_int32, c_int32, _P, _P]),
}
# entry points declared in include/nsg.h: exactly the product library's exports (tests/test_abi.py checks both directions)
HEADER_SYMBOLS = list(_SIGS)

_lib = None


class NsgError(RuntimeError):
    pass


def _bind(lib):
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = c_int32 if res is None else res
    if lib.nsg_version() != NSG_VERSION:
        raise NsgError(f"the HIP library reports ABI version {lib.nsg_version()}, this package binds version {NSG_VERSION}: rebuild it "
                       "(python -m neural_sound_generation_amd.build --force)")
    return lib


DIAG_LIB_PATH = os.path.join(_PKG, "libnsg_diag.so")
# run-time switches of the DIAGNOSTICS library (never of libnsg.so): name -> argument type
_DIAG_SWITCHES = {"nsg_debug_set_patch_gemm": c_int32, "nsg_debug_set_patch_grid": c_int32, "nsg_debug_set_wgrad_strip": c_int32,
                  "nsg_debug_set_c1_moments": c_int32, "nsg_debug_set_wgrad_bf16_native": c_int32, "nsg_debug_set_wgrad_stagger": c_int32,
                  "nsg_debug_set_wgrad_diag": c_int32, "nsg_debug_set_wgrad_stamp_buffer": c_void_p, "nsg_debug_set_stamp_buffer": c_void_p}
_diag = None


def load_diag():
    """The diagnostics build of the same library (libnsg_diag.so: -DNSG_DIAG + diag.hip; `python -m
    neural_sound_generation_amd.build --diag`): kernel-variant switches and cycle stamps for scripts/ and the A/B tests."""
    global _diag
    if _diag is None:
        import torch  # noqa: F401
        if not os.path.exists(DIAG_LIB_PATH):
            raise NsgError(f"{DIAG_LIB_PATH} is missing: python -m neural_sound_generation_amd.build --diag")
        lib = _bind(ctypes.CDLL(DIAG_LIB_PATH))
        for name, at in _DIAG_SWITCHES.items():
            fn = getattr(lib, name)
            fn.argtypes = [at]
            fn.restype = None
        _diag = lib
    return _diag


class use_diag:
    """with use_diag() as lib: every ops.* call inside goes to the diagnostics library (whose nsg_debug_set_* switches `lib`
    exposes); the product library is back afterwards."""

    def __enter__(self):
        global _lib
        load()
        self._saved = _lib
        _lib = load_diag()
        return _lib

    def __exit__(self, *exc):
        global _lib
        _lib = self._saved
        return False


def load():
    """Load libnsg.so; raises RuntimeError (never falls back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    # Device memory and streams come from PyTorch-ROCm, so libnsg.so must bind to the HIP runtime
    # torch carries (same SONAME as /opt/rocm's): load torch's first, or the process ends up with
    # two HIP runtimes and the second one finds no device.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise NsgError(
            f"{LIB_PATH} is missing: build the HIP kernels first (python -m neural_sound_generation_amd.build). "
            "There is no CPU fallback for this path.")
    _lib = _bind(ctypes.CDLL(LIB_PATH))
    return _lib


# Optional per-call census for bench.py's roofline table: when CENSUS is set (an object with begin() / end(label, start,
# flops, nbytes)), every entry-point call is bracketed by events on the launch stream.  A wrapper that knows the call's
# algorithmic work announces it with tag(label, flops, nbytes) right before the call; untagged calls go under their C name.
CENSUS = None
_TAG = None


def tag(label: str, flops: float = 0.0, nbytes: float = 0.0):
    global _TAG
    if CENSUS is not None:
        _TAG = (label, float(flops), float(nbytes))


def call(name: str, *args):
    """Call a status-returning entry point; raise with the library's message on failure."""
    global _TAG
    lib = load()
    if CENSUS is not None:
        t, _TAG = _TAG, None
        start = CENSUS.begin()
        rc = getattr(lib, name)(*args)
        label, fl, nb = t if t is not None else (name, 0.0, 0.0)
        CENSUS.end(label, start, fl, nb)
    else:
        rc = getattr(lib, name)(*args)
    if rc != 0:
        msg = lib.nsg_last_error_string()
        raise NsgError(f"{name} failed (status {rc}): {msg.decode() if msg else ''}")


def query(name: str, *args):
    return getattr(load(), name)(*args)
