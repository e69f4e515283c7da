#!/bin/bash
# c3 with bond pairs: overlapped pack pipeline (side stream, non-temporal pack with a small grid) against the serial one
for cfg in "serial 0" "overlap 192" "overlap 256" "overlap 320" "overlap 384" "serial 0" "overlap 256"; do
  set -- $cfg
  echo "== c3 pairs pack $1 workgroups $2"
  AGGF_GRAM_PACK=$1 AGGF_GRAM_PACK_WGS=$2 python bench.py --workload c3 --variant pairs --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print(round(d['ms_per_step'], 1), {k: round(v, 1) for k, v in d['config']['stage_ms_per_step'].items()}, 'frac', round(d['roofline']['frac'], 4))
"
done
