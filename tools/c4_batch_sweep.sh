#!/bin/bash
# c4 with different groupings of the sites for the batched solve (qp/gbfeat.py _solve_batches)
set -e
for cfg in "64 0.0" "32 0.85" "16 0.85" "8 0.85" "8 0.9"; do
  set -- $cfg
  echo "== min_sites $1 ratio $2"
  AGGF_FEAT_BATCH_MIN=$1 AGGF_FEAT_BATCH_RATIO=$2 python bench.py --workload c4 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print(round(d['ms_per_step'], 1), {k: round(v, 1) for k, v in d['config']['stage_ms_per_step'].items()})
print(d['roofline']['launches'][150:300])
"
done
