#!/usr/bin/env python
"""Summarise rocprofv3 output into profiles/: per-kernel time from a --kernel-trace --stats run and
per-kernel HBM traffic from two separate --pmc passes (FETCH_SIZE, WRITE_SIZE), corrected as
/opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950: bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024
(FETCH_SIZE reports half of a wide coalesced read; checked on bn_apply / relu_bwd_add whose byte counts
are known).

    python scripts/pmc_summary.py <stats_dir> <fetch_dir> <write_dir> <tag> <dtype: f32|bf16>

Writes profiles/<tag>_kernel_stats.csv, profiles/<tag>_pmc_hbm_traffic.csv and updates the <dtype> entry of
profiles/pmc_traffic_latest.json (what bench.py reports as roofline.traffic).
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").replace("unsigned short", "bf16").split("(")[0]


def find(d, suffix):
    hits = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    if not hits:
        raise SystemExit(f"no *{suffix} under {d}")
    return hits[0]


def counters(d, cname):
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(find(d, "counter_collection.csv"))):
        if r["Counter_Name"] == cname:
            agg.setdefault(short(r["Kernel_Name"]), []).append(float(r["Counter_Value"]))
    return agg


def main():
    stats_dir, fetch_dir, write_dir, tag, dtype = sys.argv[1:6]
    out_dir = os.path.join(ROOT, "profiles")
    os.makedirs(out_dir, exist_ok=True)
    rows = list(csv.DictReader(open(find(stats_dir, "kernel_stats.csv"))))
    with open(os.path.join(out_dir, f"{tag}_kernel_stats.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_ms", "avg_us", "percent"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], f"{float(r['TotalDurationNs']) / 1e6:.3f}", f"{float(r['AverageNs']) / 1e3:.1f}", r["Percentage"]])
    fa, wa = counters(fetch_dir, "FETCH_SIZE"), counters(write_dir, "WRITE_SIZE")
    table = {}
    with open(os.path.join(out_dir, f"{tag}_pmc_hbm_traffic.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "launches", "FETCH_SIZE_KB_avg", "WRITE_SIZE_KB_avg", "hbm_MB_per_launch=(2*FETCH+WRITE)*1024/1e6"])
        for k, v in fa.items():
            fe = sum(v) / len(v)
            wr = sum(wa.get(k, [0.0])) / max(1, len(wa.get(k, [0.0])))
            mb = (2 * fe + wr) * 1024 / 1e6
            table[k] = {"launches": len(v), "hbm_bytes_per_launch": (2 * fe + wr) * 1024}
            w.writerow([k, len(v), f"{fe:.1f}", f"{wr:.1f}", f"{mb:.1f}"])
    # dominant kernel: the 128x128 gather_gemm instantiations of this dtype (the launches bench.py times), launch-weighted
    # (bf16: the patch-staged kernel of gemm_patch.hip runs every 3x3 / 4x4 conv forward and data gradient of the step)
    el = "bf16, bf16" if dtype == "bf16" else "float, float"
    gg = {k: v for k, v in table.items() if k.startswith(f"gather_gemm_kernel<{el}, 2, 2, 2, 2") or (dtype == "bf16" and k.startswith("patch_gemm_kernel<"))}
    n = sum(v["launches"] for v in gg.values())
    avg = sum(v["launches"] * v["hbm_bytes_per_launch"] for v in gg.values()) / max(1, n)
    latest = os.path.join(out_dir, "pmc_traffic_latest.json")
    cur = {}
    if os.path.exists(latest):
        cur = json.load(open(latest))
        if "hbm_bytes_per_launch" in cur:      # old single-entry layout
            cur = {}
    cur[dtype] = {"tag": tag, "kernel": "patch_gemm_kernel<...> (gemm_patch.hip)" if dtype == "bf16" else f"gather_gemm_kernel<{el}, 128x128>", "launches": n, "hbm_bytes_per_launch": avg,
                  "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes of bench.py --dtype %s; "
                          "bytes=(2*FETCH_SIZE+WRITE_SIZE)*1024" % dtype}
    json.dump(cur, open(latest, "w"), indent=1)
    print("gather_gemm (%s): %.1f MB HBM traffic per launch over %d launches" % (dtype, avg / 1e6, n))


if __name__ == "__main__":
    main()
