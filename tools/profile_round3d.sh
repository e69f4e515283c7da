#!/bin/bash
# runs on the GPU box: c5 re-profiled after the conditional-mean cache and the one-workgroup tile table, plus its bench line
set -o pipefail
PASSES="fetch mfma" bash tools/profile_gpu.sh r03_c5 --workload c5 --steps 5 --warmup 2 > gpurun_out/r03_c5.log 2>&1; tail -1 gpurun_out/r03_c5.log
timeout -k 10 600 python bench.py --workload c5 > gpurun_out/r03_bench_c5.json 2> gpurun_out/r03_bench_c5.err || echo "bench c5 failed"
tail -c 200 gpurun_out/r03_bench_c5.json; echo
