#!/bin/bash
# Same-box A/B of kernel variants inside the training step: alternates bench.py runs between library builds.
#   bash scripts/ab_variants.sh <rounds> <name> [<name> ...]      name = "base" (the in-tree libnsg.so) or _exp/libnsg_<name>.so
# (build variants with scripts/build_variant.py).  Prints ms/step and the dominant kernel's in-region rate per run.
set -e
cd "$(dirname "$0")/.."
ROUNDS=$1; shift
PKG=neural_sound_generation_amd
cp $PKG/libnsg.so /tmp/libnsg_base.so
ARGS=${AB_ARGS:---no-cpu-baseline --no-second-mode --no-other-configs}
for r in $(seq 1 $ROUNDS); do
  for v in "$@"; do
    if [ "$v" = base ]; then cp /tmp/libnsg_base.so $PKG/libnsg.so; else cp _exp/libnsg_$v.so $PKG/libnsg.so; fi
    python bench.py $ARGS 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=l.get('roofline') or {}
print('round $r %-12s %8.3f ms/step  %7.1f TF in-region (%s launches, %.4f ms avg)' % ('$v', l['ms_per_step'], r.get('achieved',0), r.get('launches'), r.get('avg_launch_ms',0)))"
  done
done
cp /tmp/libnsg_base.so $PKG/libnsg.so
