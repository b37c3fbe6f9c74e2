#!/bin/bash
# rocprofv3 kernel statistics of a short bf16 bench run; prints the rows matching $1 (egrep pattern).  Run on the GPU box:
#   bash scripts/prof_kernels.sh 'c1m|bnrelu'
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pk
rocprofv3 --kernel-trace --output-format csv -d /tmp/pk -o k -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-second-mode --no-other-configs --no-kernel-timer > /tmp/pk.log 2>&1
f=$(find /tmp/pk -name "*kernel_trace.csv" | head -1)
python3 - "$f" "$1" <<'PY'
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
pat = re.compile(sys.argv[2])
acc = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if pat.search(n):
        acc[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    v2 = v[len(v) // 2:]            # the later half: captured-graph replays
    print(f"{n[:110]:110s} n={len(v):4d} avg={sum(v2) / len(v2):8.1f} us min={min(v2):8.1f}")
PY
