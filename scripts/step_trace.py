"""Per-dispatch listing of ONE training step from a rocprofv3 --kernel-trace CSV (the step between the last two
adam_kernel dispatches, or the last such step containing a kernel whose name has the optional substring -- e.g.
"gather_gemm_kernel<unsigned short" for the bf16 step of a bench run that also times fp32): start offset, duration, kernel,
blocks.  Usage: python scripts/step_trace.py <kernel_trace.csv> [substring]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
a, b = idx[-2] + 1, idx[-1] + 1
if len(sys.argv) > 2:
    for k in range(len(idx) - 1, 0, -1):
        if any(sys.argv[2] in r["Kernel_Name"] for r in rows[idx[k - 1] + 1:idx[k] + 1]):
            a, b = idx[k - 1] + 1, idx[k] + 1
            break
t0 = int(rows[a]["Start_Timestamp"])
tot = 0.0
agg = {}
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    d = (e - s) / 1e3
    tot += d
    n = r["Kernel_Name"].replace("unsigned short", "bf16").replace("(anonymous namespace)::", "").replace("void ", "")
    short = n.split("(")[0]
    agg.setdefault(short, [0, 0.0])
    agg[short][0] += 1
    agg[short][1] += d
    print(f"{(s - t0) / 1e3:9.1f} {d:8.1f}  {n[:80]:80s} blocks={int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])}x{int(r['Grid_Size_Y'])}")
print(f"sum {tot:.1f} us over {b - a} dispatches; span {(int(rows[b - 1]['End_Timestamp']) - t0) / 1e3:.1f} us")
print("---- by kernel ----")
for k, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{d:9.1f} us {100 * d / tot:5.1f}%  x{c:3d}  {k[:90]}")
