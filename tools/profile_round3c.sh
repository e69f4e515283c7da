#!/bin/bash
# runs on the GPU box: c4 re-profiled after the size-grouped solve batches and the structured A'A, plus its bench line
set -o pipefail
PASSES="fetch mfma" bash tools/profile_gpu.sh r03_c4 --workload c4 --steps 3 --warmup 1 > gpurun_out/r03_c4.log 2>&1; tail -1 gpurun_out/r03_c4.log
timeout -k 10 600 python bench.py --workload c4 --steps 5 --warmup 2 > gpurun_out/r03_bench_c4.json 2> gpurun_out/r03_bench_c4.err || echo "bench c4 failed"
tail -c 300 gpurun_out/r03_bench_c4.json; echo
