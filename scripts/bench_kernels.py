"""Print the per-kernel rows of a bench.py JSON line compactly:  python scripts/bench_kernels.py FILE [substring ...]"""
import json, sys
l = json.loads([x for x in open(sys.argv[1]).read().splitlines() if x.startswith("{")][-1])
subs = sys.argv[2:]
print("value %.0f  ms/step %.3f  gather(in-region) %.1f TF frac %.4f  step_frac %.4f" % (
    l["value"], l["ms_per_step"], l["roofline"]["achieved"], l["roofline"]["frac"], l["roofline"].get("step_frac", 0)))
for r in l["roofline"].get("kernels", []):
    if subs and not any(s in r["kernel"] for s in subs):
        continue
    print("  %-58s x%-4g %8.1f us  %5.1f%%  %8s %s" % (r["kernel"], r["per_step"], r["us_per_launch"], 100 * r["share_of_step"],
                                                       r["achieved"], r["unit"] or ""))
if l.get("other_mode"):
    o = l["other_mode"]
    print("other mode %s: value %.0f ms/step %.3f gather %.1f" % (o["dtype"], o["value"], o["ms_per_step"], o.get("roofline", {}).get("achieved", 0)))
for oc in l.get("other_configs") or []:
    print("config3 %s: value %.0f ms/step %.3f step_frac %.4f vq %s" % (oc["dtype"], oc["value"], oc["ms_per_step"], oc["step_frac"],
          (oc.get("vq_forward") or {}).get("us_per_launch")))
