#!/bin/bash
# runs on the GPU box: re-profile the workloads whose code changed after tools/profile_round3.sh, and one default
# bench line (with the CPU baseline) per workload
set -o pipefail
PASSES="fetch mfma" bash tools/profile_gpu.sh r03_c5 --workload c5 --steps 5 --warmup 2 > gpurun_out/r03_c5.log 2>&1; tail -1 gpurun_out/r03_c5.log
PASSES="fetch mfma" bash tools/profile_gpu.sh r03_c4 --workload c4 --steps 3 --warmup 1 > gpurun_out/r03_c4.log 2>&1; tail -1 gpurun_out/r03_c4.log
for w in c3 c2 c1 c4 c5; do
  timeout -k 10 600 python bench.py --workload $w > gpurun_out/r03_bench_$w.json 2> gpurun_out/r03_bench_$w.err || echo "bench $w failed"
  tail -c 300 gpurun_out/r03_bench_$w.json; echo
done
for v in pairs zeronet dense; do
  timeout -k 10 600 python bench.py --workload c3 --variant $v > gpurun_out/r03_bench_c3_$v.json 2> gpurun_out/r03_bench_c3_$v.err || echo "bench $v failed"
done
