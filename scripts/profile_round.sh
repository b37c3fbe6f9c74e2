#!/bin/bash
# rocprofv3 passes of bench.py for both compute modes: kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in their
# own --pmc passes (never combined with a trace domain other than --kernel-trace).  Run on the GPU box from the repo root:
#   bash scripts/profile_round.sh <out-dir under gpurun_out>
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for DT in bf16 f32; do
  # the timing pass runs bench.py's own default length (20 + 5 steps): its per-kernel averages are comparable with the
  # live HIP-event timer of a plain run; the counter passes only need a few steps
  SARGS="--dtype $DT --steps 20 --warmup 5 --no-cpu-baseline --no-second-mode --no-other-configs --no-kernel-timer"
  ARGS="--dtype $DT --steps 4 --warmup 2 --no-cpu-baseline --no-second-mode --no-other-configs --no-kernel-timer"
  rocprofv3 --kernel-trace --stats -d $OUT/${DT}_stats -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py $SARGS > $OUT/${DT}_stats.log 2>&1
  echo "stats $DT done"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/${DT}_fetch -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/${DT}_fetch.log 2>&1
  echo "fetch $DT done"
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/${DT}_write -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/${DT}_write.log 2>&1
  echo "write $DT done"
done
