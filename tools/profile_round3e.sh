#!/bin/bash
# runs on the GPU box: c3 + bond pairs re-profiled with the overlapped pack pipeline, plus its bench line
set -o pipefail
PASSES="fetch mfma" bash tools/profile_gpu.sh r03_c3pairs --workload c3 --variant pairs --steps 3 --warmup 1 > gpurun_out/r03_c3pairs.log 2>&1; tail -1 gpurun_out/r03_c3pairs.log
timeout -k 10 600 python bench.py --workload c3 --variant pairs > gpurun_out/r03_bench_c3_pairs.json 2> gpurun_out/r03_bench_c3_pairs.err || echo "bench failed"
tail -c 200 gpurun_out/r03_bench_c3_pairs.json; echo
