"""Condense rocprofv3 output (gpurun_out/<tag>_trace|_fetch|_write|_dram) into
gpurun_out/<tag>_summary.json.  Runs on the GPU box right after tools/profile_gpu.sh."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

tag = sys.argv[1]
out_dir = "gpurun_out"


def find(sub, pattern):
    hits = glob.glob(os.path.join(out_dir, f"{tag}_{sub}", "**", pattern), recursive=True)
    return hits[0] if hits else None


summary = {"tag": tag}


def source_sha1():
    """Hash of the sources the profiled command ran from (this scratch copy has no .git): tools/commit_profile.py
    recomputes it in the repository and records the commit only if the two agree."""
    import hashlib

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha1()
    files = []
    for pat in ("aggforce_amd/csrc/*.hip", "aggforce_amd/csrc/*.h", "include/*.h", "aggforce_amd/*.py", "aggforce_amd/*/*.py", "bench.py"):
        files += glob.glob(os.path.join(root, pat))
    for f in sorted(files):
        h.update(os.path.relpath(f, root).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


summary["source_sha1"] = source_sha1()
stats = find("trace", "*kernel_stats.csv")
if stats:
    rows = list(csv.DictReader(open(stats)))
    keep = [{"kernel": r["Name"][:110], "calls": int(r["Calls"]), "total_ms": float(r["TotalDurationNs"]) / 1e6,
             "avg_ms": float(r["AverageNs"]) / 1e6, "pct": float(r["Percentage"])} for r in rows]
    keep.sort(key=lambda x: -x["total_ms"])
    summary["kernel_stats"] = keep[:25]
for name in ("fetch", "write", "dram", "mfma", "insts"):
    f = find(name, "*counter_collection.csv")
    if not f:
        continue
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for r in csv.DictReader(open(f)):
        a = acc[r["Kernel_Name"][:110]][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"])
        a[1] += 1
    top = sorted(acc.items(), key=lambda kv: -max(v[0] for v in kv[1].values()))[:8]
    summary[f"pmc_{name}"] = [
        {"kernel": k, **{c: {"dispatches": v[1], "per_dispatch": v[0] / v[1]} for c, v in cs.items()}} for k, cs in top
    ]
json.dump(summary, open(os.path.join(out_dir, f"{tag}_summary.json"), "w"), indent=1)
g = [k for k in summary.get("kernel_stats", []) if "gram_tile" in k["kernel"]]
print("gram kernel:", g[:2])
for name in ("fetch", "write", "dram", "mfma"):
    for e in summary.get(f"pmc_{name}", [])[:3]:
        print(name, json.dumps(e)[:400])
