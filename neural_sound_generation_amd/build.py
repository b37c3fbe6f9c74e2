"""Builds libnsg.so (hand-written gfx950 HIP kernels behind the C ABI of include/nsg.h) in-tree.

    python -m neural_sound_generation_amd.build [--force]

hipcc cross-compiles for gfx950 without a GPU.  No torch headers, no vendor math libraries: the
library links only the HIP runtime.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(PKG, "libnsg.so")
# The diagnostics library (scripts/ and the kernel-vs-kernel A/B tests): the same sources under -DNSG_DIAG -- run-time switches
# between kernel variants (nsg_debug_set_*), in-kernel cycle stamps -- plus the probe kernels of diag.hip.  None of that is in
# libnsg.so: the product library exports exactly what include/nsg.h declares.
OBJ_DIAG = os.path.join(CSRC, "_obj_diag")
LIB_DIAG = os.path.join(PKG, "libnsg_diag.so")
SOURCES = ["api_common.hip", "gemm_gather.hip", "gemm_patch.hip", "gemm_wgrad.hip", "gemm_wgrad_strip.hip", "gemm_flat.hip", "vq.hip", "segsum.hip", "vq_bf16.hip", "bn.hip", "elementwise.hip", "conv_api.hip", "stencil_c1.hip", "c1_mfma.hip", "prior_ops.hip", "audio.hip"]
DIAG_SOURCES = SOURCES + ["diag.hip"]
# -ffp-contract=off: the bit-exact VQ path spells out every fma itself; nothing may be re-fused.
FLAGS = FLAGS_ = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-fvisibility=hidden",
         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-Wall", "-Wno-unused-function"]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm with gfx950 support)")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True, diag: bool = False) -> str:
    """Build libnsg.so (diag=False) or libnsg_diag.so (diag=True); returns the library's path."""
    OBJ, LIB, SOURCES, FLAGS = ((OBJ_DIAG, LIB_DIAG, DIAG_SOURCES, FLAGS_ + ["-DNSG_DIAG"]) if diag else
                                (globals()["OBJ"], globals()["LIB"], globals()["SOURCES"], FLAGS_))
    os.makedirs(OBJ, exist_ok=True)
    # every header any source may include: editing one rebuilds all objects (seconds per file)
    headers = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + [os.path.join(ROOT, "include", "nsg.h")]
    cc = hipcc()
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + headers):
            jobs.append((s, o))

    def compile_one(job):
        s, o = job
        cmd = [cc] + FLAGS + ["-c", s, "-o", o]
        if verbose:
            print("[nsg build]" + (" (diag)" if diag else ""), os.path.basename(s), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        list(ex.map(compile_one, jobs))
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        subprocess.check_call([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
        if verbose:
            print("[nsg build] linked", LIB, flush=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    if "--diag" in sys.argv:
        build(force="--force" in sys.argv, diag=True)
