"""Condense rocprofv3 output (gpurun_out/<tag>_trace|_fetch|_write) into profiles/<tag>_*.  Runs
on the GPU box right after tools/profile_gpu.sh, or locally on merged gpurun_out/."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

tag = sys.argv[1]
out_dir = "gpurun_out"
os.makedirs(out_dir, exist_ok=True)


def find(sub, pattern):
    hits = glob.glob(os.path.join(out_dir, f"{tag}_{sub}", "**", pattern), recursive=True)
    return hits[0] if hits else None


summary = {"tag": tag}
stats = find("trace", "*kernel_stats.csv")
if stats:
    rows = list(csv.DictReader(open(stats)))
    keep = []
    for r in rows:
        keep.append({"kernel": r["Name"][:110], "calls": int(r["Calls"]), "total_ms": float(r["TotalDurationNs"]) / 1e6,
                     "avg_ms": float(r["AverageNs"]) / 1e6, "pct": float(r["Percentage"])})
    keep.sort(key=lambda x: -x["total_ms"])
    summary["kernel_stats"] = keep[:25]
for name in ("fetch", "write"):
    f = find(name, "*counter_collection.csv")
    if not f:
        continue
    acc = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:110]
        acc[k][0] += float(r["Counter_Value"])
        acc[k][1] += 1
    top = sorted(acc.items(), key=lambda kv: -kv[1][0])[:12]
    # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB
    summary[f"pmc_{name}_KiB"] = [{"kernel": k, "dispatches": v[1], "sum": v[0], "per_dispatch": v[0] / v[1]} for k, v in top]
json.dump(summary, open(os.path.join(out_dir, f"{tag}_summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1)[:6000])
