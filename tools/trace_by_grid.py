"""Condense a rocprofv3 --kernel-trace CSV: per (kernel, grid) the launch count, total and average duration -- for the
launch chains of K2, whose one kernel template runs at many shapes.  Usage: trace_by_grid.py <dir> [name filter...]"""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
filt = sys.argv[2:]
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
acc = defaultdict(lambda: [0, 0.0])
t_first, t_last = None, None
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"]
    if filt and not any(x in name for x in filt):
        continue
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    grid = (r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""),
            r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")))
    a = acc[(name[:90], grid)]
    a[0] += 1
    a[1] += dur
tot = sum(v[1] for v in acc.values())
print(f"{len(acc)} (kernel, grid) groups, {sum(v[0] for v in acc.values())} launches, {tot / 1e3:.2f} ms of kernel time")
for (name, grid), (n, us) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{us / 1e3:9.3f} ms {n:6d} x {us / n:9.1f} us  grid {'x'.join(g for g in grid[:3] if g)} wg {grid[3]}  {name}")
