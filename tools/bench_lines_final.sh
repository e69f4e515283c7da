#!/bin/bash
# runs on the GPU box: one default bench line (with the CPU baseline) per workload and variant, from the final tree
set -o pipefail
for w in c3 c2 c1 c4 c5; do
  timeout -k 10 600 python bench.py --workload $w > gpurun_out/r03_bench_$w.json 2> gpurun_out/r03_bench_$w.err || echo "bench $w failed"
  tail -c 200 gpurun_out/r03_bench_$w.json; echo
done
for v in pairs zeronet dense; do
  timeout -k 10 600 python bench.py --workload c3 --variant $v > gpurun_out/r03_bench_c3_$v.json 2> gpurun_out/r03_bench_c3_$v.err || echo "bench $v failed"
done
