"""Looks through the gfx950 assembly of every kernel for small rolled loops that load from global memory and wait with
s_waitcnt vmcnt(0) inside the loop: each iteration is then one full memory round trip (the flat 1x1 GEMM's weight prologue
spent 20-40 us per block that way).  Usage: python scripts/scan_serial_loops.py  (compiles csrc/*.hip to /tmp/asm first)."""
import glob, os, re, subprocess, sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "neural_sound_generation_amd", "csrc")
OUT = "/tmp/asm"


def compile_one(src):
    dst = os.path.join(OUT, os.path.basename(src)[:-4] + ".s")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fvisibility=hidden",
                    "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only", src, "-o", dst], stderr=subprocess.DEVNULL, check=True)
    return dst


def demangle(name):
    try:
        return subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    except OSError:
        return name


def main():
    os.makedirs(OUT, exist_ok=True)
    if "--no-compile" not in sys.argv:
        with ThreadPoolExecutor(4) as ex:
            list(ex.map(compile_one, sorted(glob.glob(os.path.join(CSRC, "*.hip")))))
    for fn in sorted(glob.glob(os.path.join(OUT, "*.s"))):
        lines = open(fn).read().split("\n")
        func, labels = None, {}
        waits = {}
        for i, l in enumerate(lines):
            m = re.match(r"^(_Z\w+):", l)
            if m:
                func, labels = m.group(1), {}
            if func and "vmcnt(0)" in l:
                waits[func] = waits.get(func, 0) + 1
        for f, k in waits.items():
            if k >= 12:      # straight-line code can serialise too: loads under lane conditions each get a branch and a full wait
                print(f"{os.path.basename(fn):22s} {demangle(f)[:100]:100s} {k} x s_waitcnt vmcnt(0) in the kernel")
        func = None
        for i, l in enumerate(lines):
            m = re.match(r"^(_Z\w+):", l)
            if m:
                func, labels = m.group(1), {}
            m = re.match(r"^(\.LBB\d+_\d+):", l)
            if m:
                labels[m.group(1)] = i
            m = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l)
            if m and m.group(1) in labels:
                body = lines[labels[m.group(1)]:i]
                nl = sum(1 for b in body if re.search(r"(global|buffer)_load", b))
                w0 = sum(1 for b in body if "vmcnt(0)" in b)
                ninst = sum(1 for b in body if b.startswith("\t") and not b.startswith("\t;"))
                if nl and w0 and ninst < 80:
                    print(f"{os.path.basename(fn):22s} {demangle(func)[:100]:100s} loop of {ninst} instructions, {nl} loads, {w0} vmcnt(0)")


if __name__ == "__main__":
    main()
