#!/bin/bash
# The launch sequence of one captured bf16 step (kernel name, duration, gap to the previous kernel): gpurun_out/step_sequence.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sq
rocprofv3 --kernel-trace --output-format csv -d /tmp/sq -o k -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-second-mode --no-other-configs --no-kernel-timer > /tmp/sq.log 2>&1
f=$(find /tmp/sq -name "*kernel_trace.csv" | head -1)
python3 - "$f" > $GRAFT_REPO_ROOT/gpurun_out/step_sequence.txt <<'PY'
import csv, sys, re
rows = sorted(list(csv.DictReader(open(sys.argv[1]))), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# the last step: from the last launch of the first kernel of a step (the input layer's tap moments / pack) to the end
last = max(i for i, n in enumerate(names) if "pack_w_kernel" in n)
prev_end = None
tot = gaps = 0
for r in rows[last:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    n = re.sub(r"^void ", "", n).split("(")[0]
    print(f"{n[:70]:70s} {(e - s) / 1e3:8.1f} us  gap {gap:6.1f}")
    tot += (e - s) / 1e3; gaps += max(gap, 0.0)
    prev_end = e
print(f"launches {len(rows) - last}, kernel time {tot:.1f} us, gaps {gaps:.1f} us")
PY
tail -1 $GRAFT_REPO_ROOT/gpurun_out/step_sequence.txt
