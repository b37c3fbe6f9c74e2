#!/bin/bash
# Same-box A/B with the per-kernel table: bash scripts/ab_kernels.sh "<substr>|<substr>..." <variant> [<variant> ...]
cd "$(dirname "$0")/.."
PAT=$1; shift
PKG=neural_sound_generation_amd
cp $PKG/libnsg.so /tmp/libnsg_base.so
mkdir -p gpurun_out
for r in 1 2; do for v in "$@"; do
  if [ "$v" = base ]; then cp /tmp/libnsg_base.so $PKG/libnsg.so; else cp _exp/libnsg_$v.so $PKG/libnsg.so; fi
  python bench.py --no-cpu-baseline --no-second-mode --no-other-configs 2>/dev/null | python -c "
import json,sys,re
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
t=json.load(open('gpurun_out/bench_kernels.json'))
rows=list(t.values())[0]
sel=[r for r in rows if re.search(r'$PAT', r['kernel'])]
print('%-8s %.3f ms/step | ' % ('$v', l['ms_per_step']) + ' | '.join('%s %.1f' % (r['kernel'][:28], r['us_per_launch']) for r in sel))"
done; done
cp /tmp/libnsg_base.so $PKG/libnsg.so
