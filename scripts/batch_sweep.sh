#!/bin/bash
# bf16 configs[1] step at several clips/GPU (report the best of 32/64/128): one bench line each
for b in 32 48 64 96 128 192 256; do
  python bench.py --batch $b --steps 30 --warmup 8 --no-cpu-baseline --no-second-mode --no-other-configs --no-kernel-timer 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); print($b, d['value'], d['ms_per_step'])"
done
