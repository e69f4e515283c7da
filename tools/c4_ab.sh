#!/bin/bash
# c4: A/B of the structured A'A (AGGF_FEAT_ATA) -- alternating runs, to see the run-to-run noise as well
for i in 1 2 3; do
  for ata in 0 1; do
    echo "== ata $ata"
    AGGF_FEAT_ATA=$ata python bench.py --workload c4 --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print(round(d['ms_per_step'], 1), {k: round(v, 1) for k, v in d['config']['stage_ms_per_step'].items()})
"
  done
done
