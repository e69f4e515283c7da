"""Timeline of ONE step out of a rocprofv3 --kernel-trace CSV of bench.py: the kernels in start order with the idle gap
in front of each (all queues merged: a gap = no kernel of the process running), and the sums.
    python tools/trace_gaps.py <dir> [marker kernel substring = gram_tile_dma_kernel]"""
import csv
import glob
import os
import sys

d = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "gram_tile_dma_kernel"
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))))
starts = [i for i, r in enumerate(rows) if marker in r[2]]
# the second-to-last step: from one marker kernel to the next
i0, i1 = starts[-2], starts[-1]
# (the step begins with the small kernels in front of the marker: walk back over kernels closer than 200 us)
while i0 > 0 and rows[i0][0] - rows[i0 - 1][1] < 200_000 and marker not in rows[i0 - 1][2]:
    i0 -= 1
while i1 > i0 and rows[i1][0] - rows[i1 - 1][1] < 200_000 and marker not in rows[i1 - 1][2]:
    i1 -= 1
step = rows[i0:i1]
busy_end = step[0][0]
gap_sum = 0
for s, e, n in step:
    gap = max(0, s - busy_end)
    gap_sum += gap
    if gap > 3000 or (e - s) > 50_000:
        print(f"gap {gap / 1e3:8.1f} us | {(e - s) / 1e3:9.1f} us  {n[:100]}")
    busy_end = max(busy_end, e)
print(f"step: {len(step)} kernels, {(busy_end - step[0][0]) / 1e3:.1f} us from first start to last end, idle {gap_sum / 1e3:.1f} us; "
      f"to the next step's first kernel: {(rows[i1][0] - busy_end) / 1e3:.1f} us")
