#!/bin/bash
# GPU box: L2 / fabric counters of the Gram kernel alone (tools/gram_ablate quick mode).
export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gram_ablate.hip aggforce_amd/csrc/aggf_util.hip -o /tmp/gram_ablate 2>&1 | grep " error"
/tmp/gram_ablate quick | grep -E "LDS-DMA ring:"
rm -rf gpurun_out/pmc_gram
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pmc_gram -- /tmp/gram_ablate quick > gpurun_out/pmc_gram.log 2>&1
python3 - <<'PY'
import csv, glob
from collections import defaultdict
f = glob.glob("gpurun_out/pmc_gram/**/*counter_collection.csv", recursive=True)[0]
by = defaultdict(dict)
for r in csv.DictReader(open(f)):
    if "gram_tile" in r["Kernel_Name"]:
        by[(int(r["Dispatch_Id"]), r["Kernel_Name"][11:60])][r["Counter_Name"]] = float(r["Counter_Value"])
seen = set()
for k in sorted(by):
    if k[1] in seen: continue
    seen.add(k[1]); v = by[k]
    print(k[1], {n: f"{x:.3e}" for n, x in v.items()}, "hit rate %.2f" % (v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"])))
PY
