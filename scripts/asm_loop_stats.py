"""Instruction mix of the MFMA loops of a kernel in a hipcc -S listing.
Usage: python scripts/asm_loop_stats.py file.s <substring of the mangled kernel name> [min_mfma]"""
import collections
import re
import sys

txt = open(sys.argv[1]).read()
want = sys.argv[2]
min_mfma = int(sys.argv[3]) if len(sys.argv) > 3 else 8
for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)\n\s*s_endpgm', txt, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if want not in name:
        continue
    lines = body.split('\n')
    print(name, len(lines), "lines")
    labels = [i for i, l in enumerate(lines) if l.startswith('.LBB')]
    for a, b in zip(labels, labels[1:] + [len(lines)]):
        seg = lines[a:b]
        if sum('v_mfma' in l for l in seg) < min_mfma:
            continue
        c = collections.Counter()
        for l in seg:
            l = l.strip()
            if not l or l.startswith(';') or l.startswith('.'):
                continue
            op = l.split()[0]
            key = ('mfma' if 'mfma' in op else 'ds_read' if op.startswith('ds_read') else 'ds_write' if op.startswith('ds_write')
                   else 'vmem' if op.startswith(('buffer_', 'global_', 'flat_')) else 'waitcnt' if op == 's_waitcnt'
                   else 'barrier' if op == 's_barrier' else 'valu' if op.startswith('v_') else 'salu' if op.startswith('s_') else op)
            c[key] += 1
        print(" ", lines[a].split(':')[0], dict(c))
