#!/bin/bash
# configs[3] (D=256, K=8192) bf16 step at the clips/GPU SURVEY 8d lists (8, 16, 32) and beyond: one line each
for b in 8 16 32 48 64; do
  python bench.py --dim 256 --z-dim 8192 --batch $b --steps 20 --warmup 5 --no-cpu-baseline --no-second-mode --no-other-configs --no-kernel-timer 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); print($b, d['value'], d['ms_per_step'])"
done
