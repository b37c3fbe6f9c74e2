"""Build an experimental copy of libnsg.so with one source compiled under extra -D flags (for A/B timing of kernel variants on
one GPU box: `cp _exp/libnsg_<name>.so neural_sound_generation_amd/libnsg.so` between runs inside one gpurun command).
    python scripts/build_variant.py <name> <source.hip> [-DFLAG ...]
Writes _exp/libnsg_<name>.so (git-ignored); the regular library and its objects are left alone."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_sound_generation_amd import build as B

name, src, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
B.build(verbose=False)
out = os.path.join(B.ROOT, "_exp")
os.makedirs(out, exist_ok=True)
obj = os.path.join(out, f"{name}_{src.replace('.hip', '.o')}")
subprocess.check_call([B.hipcc()] + B.FLAGS + extra + ["-c", os.path.join(B.CSRC, src), "-o", obj])
objs = [obj if s == src else os.path.join(B.OBJ, s.replace(".hip", ".o")) for s in B.SOURCES]
lib = os.path.join(out, f"libnsg_{name}.so")
subprocess.check_call([B.hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
print(lib)
