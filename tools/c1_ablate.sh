#!/bin/bash
# phase shares of the small-system Gram kernel at CLN025 x 4e6 frames (AGGF_SMALL_ABL: 1 no MFMA, 2 no group sums, 3 no DMA)
for a in 0 1 2 3; do
  AGGF_SMALL_ABL=$a python bench.py --workload c1 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ABL $a gram ms', round(d['config']['stage_ms_per_step']['gram'],2))"
done
