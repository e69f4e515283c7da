#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel trace + separate PMC passes of bench.py.
# Usage: bash tools/profile_gpu.sh <tag> [bench args...]     outputs under gpurun_out/<tag>_*
set -o pipefail
tag=$1; shift
export TMPDIR=/tmp
out=gpurun_out
rm -rf $out/${tag}_trace $out/${tag}_fetch $out/${tag}_write
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_trace -- python3 bench.py --no-cpu-baseline "$@" > $out/${tag}_trace.log 2>&1 || { tail -5 $out/${tag}_trace.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/${tag}_fetch -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 0 "$@" > $out/${tag}_fetch.log 2>&1 || { tail -5 $out/${tag}_fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/${tag}_write -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 0 "$@" > $out/${tag}_write.log 2>&1 || { tail -5 $out/${tag}_write.log; exit 1; }
python3 tools/summarize_profile.py $tag
