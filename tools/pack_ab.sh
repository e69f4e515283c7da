#!/bin/bash
# c3 with bond pairs: overlapped pack pipeline (side stream, one non-temporal pack workgroup per CU) against the serial
# one and against the same chunks on one stream (AGGF_GRAM_PACK, read on every call)
for form in serial overlap chunked serial overlap; do
  echo "== c3 pairs pack $form"
  AGGF_GRAM_PACK=$form python bench.py --workload c3 --variant pairs --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print(round(d['ms_per_step'], 1), {k: round(v, 1) for k, v in d['config']['stage_ms_per_step'].items()}, 'frac', round(d['roofline']['frac'], 4))
"
done
