"""Static scan of the compiled kernels for MFMAs issued right behind a FULL LDS wait (s_waitcnt lgkmcnt(0)): the signature of hipcc
giving successive operand reads the same destination registers, so that every MFMA group waits out a whole LDS round trip (what
held the exact search at 0.45-0.6 of its peak).  usage: python scripts/scan_mfma_waits.py [file.hip ...]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neural_sound_generation_amd import build as B
files = sys.argv[1:] or [s for s in B.SOURCES if s not in ("api_common.hip", "elementwise.hip", "audio.hip", "prior_ops.hip", "segsum.hip")]
for f in files:
    asm = f"/tmp/scan_{f}.s"
    subprocess.check_call([B.hipcc()] + B.FLAGS + ["-S", "--cuda-device-only", "-o", asm, os.path.join(B.CSRC, f)], stderr=subprocess.DEVNULL)
    txt = open(asm).read()
    for m in re.finditer(r'^(_Z\w+):.*?\.amdhsa_kernel', txt, re.S | re.M):
        name, body = m.group(1), m.group(0).split('\n')
        ins = [l.strip() for l in body if l.strip() and not l.strip().startswith((';', '.'))]
        mf = sum(l.startswith('v_mfma') for l in ins)
        if mf < 8:
            continue
        hit = sum(1 for i, l in enumerate(ins) if l.startswith('v_mfma') and any(re.match(r's_waitcnt.*lgkmcnt\(0\)', p) for p in ins[max(0, i - 2):i]))
        dn = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "")[:100]
        print(f"{100 * hit // mf:3d} %  {hit:4d} / {mf:4d} MFMAs right behind lgkmcnt(0)   {f}: {dn}")
