"""CPU: libnsg.so loads and exports every entry point include/nsg.h declares; argument validation
and the pure size queries work without a GPU (no kernel is launched here)."""
import ctypes
import os
import re

import pytest

from neural_sound_generation_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "nsg.h")).read()
    return sorted(set(re.findall(r"NSG_API\s+[\w\s\*]+?\b(nsg_\w+)\s*\(", text)))


def exported_symbols(path):
    """Dynamic symbols a shared object DEFINES (nm -D --defined-only), nsg_* only."""
    import subprocess
    out = subprocess.check_output(["nm", "-D", "--defined-only", path], text=True)
    return sorted({ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-1].startswith("nsg_")})


def test_library_exports_exactly_the_declared_symbols():
    """include/nsg.h IS the ABI, in both directions: every declared entry point is exported, and libnsg.so exports no nsg_*
    symbol the header does not declare (round 2's library carried 13 undeclared nsg_debug_set_* switches over process-global
    state; they now exist only in the diagnostics build, libnsg_diag.so)."""
    lib = _lib.load()
    declared = header_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), f"{name} is declared in include/nsg.h but not exported by libnsg.so"
    assert sorted(_lib.HEADER_SYMBOLS) == declared, "the ctypes table and include/nsg.h disagree"
    exported = exported_symbols(_lib.LIB_PATH)
    extra = sorted(set(exported) - set(declared))
    assert not extra, f"libnsg.so exports symbols include/nsg.h does not declare: {extra}"
    assert not [n for n in exported if n.startswith("nsg_debug_set")], "a run-time variant switch leaked into the product library"
    m = re.search(r"#define\s+NSG_VERSION\s+(\d+)", open(os.path.join(ROOT, "include", "nsg.h")).read())
    assert lib.nsg_version() == int(m.group(1)) == _lib.NSG_VERSION


def test_diagnostics_library_is_a_superset_with_the_switches():
    if not os.path.exists(_lib.DIAG_LIB_PATH):
        pytest.skip("libnsg_diag.so not built (python -m neural_sound_generation_amd.build --diag)")
    exported = set(exported_symbols(_lib.DIAG_LIB_PATH))
    assert set(header_symbols()) <= exported
    assert set(_lib._DIAG_SWITCHES) <= exported


def test_invalid_arguments_are_rejected_before_any_launch():
    lib = _lib.load()
    null = ctypes.c_void_p(0)
    assert lib.nsg_rowsumsq(null, 4, 8, null, null) == -1
    assert b"nsg_rowsumsq" in lib.nsg_last_error_string()
    with pytest.raises(_lib.NsgError, match="nsg_adam_step"):
        _lib.call("nsg_adam_step", null, null, null, null, 4, 1e-3, 0.9, 0.999, 1e-8, 1, 1.0, null)
    # D > 256 is outside what the VQ kernel implements
    buf = (ctypes.c_float * 4)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert lib.nsg_vq_forward(p, p, 1, 300, 4, p, null, null, p, 1 << 20, null) == -2


def test_size_queries():
    lib = _lib.load()
    assert lib.nsg_vq_workspace_bytes(1000, 128, 512) >= (1000 + 512) * 4
    d = _lib.ConvDesc(2, 20, 16, 8, 20, 16, 8, 3, 1, 1, 0)
    assert lib.nsg_packed_weight_floats(ctypes.byref(d)) == 9 * 8 * 8
    assert lib.nsg_conv_workspace_bytes(ctypes.byref(d)) > 0
    bad = _lib.ConvDesc(2, 20, 16, 6, 20, 16, 8, 3, 1, 1, 0)   # C_in % 4 != 0
    assert lib.nsg_conv_workspace_bytes(ctypes.byref(bad)) == 0
    assert lib.nsg_bn_workspace_bytes(5120, 128) > 0
    assert lib.nsg_reduce_workspace_bytes(10) >= 8


def test_packed_weight_image_sizes_include_the_fragment_ordered_copy():
    """nsg_conv_forward / nsg_conv_dgrad read, for the bf16 shapes gemm_patch.hip takes, a fragment-ordered copy BEHIND the plain
    [tap][n][c] image: the allocation size the header tells callers to use must account for it -- and must not
    for the shapes that have none."""
    lib = _lib.load()
    def n(desc):
        return lib.nsg_packed_weight_floats(ctypes.byref(desc))
    plain = 9 * 128 * 128
    assert n(_lib.ConvDesc(2, 20, 64, 128, 20, 64, 128, 3, 1, 1, 0, _lib.NSG_BF16)) == 2 * plain          # bf16 3x3 128->128: twin
    assert n(_lib.ConvDesc(2, 20, 64, 128, 20, 64, 128, 3, 1, 1, 0, _lib.NSG_F32)) == plain               # fp32: none
    assert n(_lib.ConvDesc(2, 20, 64, 32, 20, 64, 32, 3, 1, 1, 0, _lib.NSG_BF16)) == 9 * 32 * 32          # narrow bf16: none
    assert n(_lib.ConvDesc(2, 20, 64, 256, 40, 128, 256, 4, 2, 1, 1, _lib.NSG_BF16)) == 2 * 16 * 256 * 256   # transposed 4/2/1, D = 256


def test_gather_gemm_rejects_tensors_past_its_32_bit_row_offsets():
    """A transposed conv's output is 4x its input: (820, 20, 256, 128) bf16 passes the 4 GiB operand limit but the
    (820, 40, 512, 128) output has more than 2^31 elements, which the kernel's int row offsets cannot address
    (ADVICE r1).  The launcher must refuse before touching the GPU -- fake pointers, no allocation, runs on CPU."""
    lib = _lib.load()
    fake, null = ctypes.c_void_p(0x10000), ctypes.c_void_p(0)
    d = _lib.ConvDesc(820, 20, 256, 128, 40, 512, 128, 4, 2, 1, 1, _lib.NSG_BF16)
    assert lib.nsg_conv_forward(ctypes.byref(d), fake, fake, null, fake, 0, fake, 1 << 30, null) == -2
    assert b"2^31" in lib.nsg_last_error_string()
    # the data gradient of Conv2d(128, 128, 4, 2, 1) writes the same 4x tensor (with the fused add / mask operands on the same offsets)
    d = _lib.ConvDesc(820, 40, 512, 128, 20, 256, 128, 4, 2, 1, 0, _lib.NSG_BF16)
    assert lib.nsg_conv_dgrad_relu_add(ctypes.byref(d), fake, fake, fake, fake, fake, 0, fake, 1 << 30, null) == -2
    assert b"2^31" in lib.nsg_last_error_string()
    # one clip fewer fits: the guard must not fire early (it would launch, so only the pure checks are exercised up to here)
    d = _lib.ConvDesc(1 << 20, 20, 256, 128, 40, 512, 128, 4, 2, 1, 1, _lib.NSG_BF16)   # the row count itself overflows
    assert lib.nsg_conv_forward(ctypes.byref(d), fake, fake, null, fake, 0, fake, 1 << 30, null) == -2


def test_loader_refuses_a_library_of_another_abi_version(monkeypatch):
    """entry points changed signature under an unchanged NSG_VERSION, and a stale libnsg.so then took a stream handle
    for a data pointer.  NSG_VERSION is now bumped on any signature change and the loader compares it."""
    lib = _lib.load()
    monkeypatch.setattr(_lib, "NSG_VERSION", lib.nsg_version() + 1)
    with pytest.raises(_lib.NsgError, match="ABI version"):
        _lib._bind(ctypes.CDLL(_lib.LIB_PATH))
