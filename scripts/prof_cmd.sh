#!/bin/bash
# rocprofv3 kernel trace of an arbitrary python script; prints per-kernel avg/min of the rows matching $1:  bash scripts/prof_cmd.sh '<egrep>' script.py [args]
PAT=$1; shift
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
rm -rf /tmp/pc
rocprofv3 --kernel-trace --output-format csv -d /tmp/pc -o k -- python3 $GRAFT_REPO_ROOT/"$@" > /tmp/pc.log 2>&1
f=$(find /tmp/pc -name "*kernel_trace.csv" | head -1)
python3 - "$f" "$PAT" <<'PY'
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
pat = re.compile(sys.argv[2])
acc = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if pat.search(n):
        acc[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)
    print(f"{n[:100]:100s} n={len(v):4d} median={v2[len(v2) // 2]:8.1f} us min={v2[0]:8.1f}")
PY
