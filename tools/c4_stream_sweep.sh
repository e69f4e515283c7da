#!/bin/bash
# c4: number of streams the featurised fit deals its per-site launches over (AGGF_FEAT_STREAMS)
for n in 1 2 3 4 6 8; do
  echo "== streams $n"
  AGGF_FEAT_STREAMS=$n python bench.py --workload c4 --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print(round(d['ms_per_step'], 1), {k: round(v, 1) for k, v in d['config']['stage_ms_per_step'].items()})
"
done
