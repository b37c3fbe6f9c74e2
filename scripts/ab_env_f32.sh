#!/bin/bash
# Same-box A/B of an environment switch inside the fp32 training step: bash scripts/ab_env_f32.sh VAR v1 v2 [...]
VAR=$1; shift
for r in 1 2 3; do for v in "$@"; do
  env $VAR=$v python bench.py --dtype f32 --steps 10 --warmup 3 --no-cpu-baseline --no-second-mode --no-other-configs --no-kernel-timer 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); print('$VAR=$v', d['value'], d['ms_per_step'])"
done; done
