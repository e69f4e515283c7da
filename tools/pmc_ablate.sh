#!/bin/bash
# GPU box: SQ counters of the Gram ablation variants (one dispatch row per variant/rep).
export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gram_ablate.hip aggforce_amd/csrc/aggf_util.hip -o /tmp/gram_ablate 2>&1 | grep " error"
rm -rf gpurun_out/pmc_ablate*
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/pmc_ablate1 -- /tmp/gram_ablate quick > gpurun_out/pmc_ablate1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_MFMA --output-format csv -d gpurun_out/pmc_ablate2 -- /tmp/gram_ablate quick > gpurun_out/pmc_ablate2.log 2>&1
python3 - <<'PY'
import csv, glob
from collections import defaultdict
for d in ("pmc_ablate1", "pmc_ablate2"):
    f = glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True)
    if not f:
        print("no csv for", d); continue
    rows = list(csv.DictReader(open(f[0])))
    by = defaultdict(dict)
    for r in rows:
        if "gram_tile" not in r["Kernel_Name"]: continue
        by[(int(r["Dispatch_Id"]), r["Kernel_Name"][28:70])][r["Counter_Name"]] = float(r["Counter_Value"])
    for k in sorted(by):
        print(k, {n: f"{v:.3e}" for n, v in by[k].items()})
PY
