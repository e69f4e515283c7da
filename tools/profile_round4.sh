#!/bin/bash
# Runs ON THE GPU BOX: round-4 profiles of every workload (kernel trace + separate PMC passes) from ONE tree.
#   bash tools/profile_round4.sh a   -> c3 and its variants          (profiles/r04_c3*)
#   bash tools/profile_round4.sh b   -> c1, c2, c4, c5               (profiles/r04_c{1,2,4,5}*)
#   bash tools/profile_round4.sh c   -> bench lines with the CPU baseline, stream kernels, next rows, small-system ablations
# Afterwards, in the build container:  for t in ...; do python tools/commit_profile.py r04_$t "<bench args>"; done
set -o pipefail
part=${1:-a}
run() { tag=$1; shift; bash tools/profile_gpu.sh $tag "$@" > gpurun_out/$tag.log 2>&1; tail -1 gpurun_out/$tag.log; }
if [ "$part" = a ]; then
  PASSES="fetch write dram mfma" run r04_c3 --workload c3 --steps 3 --warmup 1
  PASSES="fetch mfma" run r04_c3pairs --workload c3 --variant pairs --steps 3 --warmup 1
  PASSES="trace" run r04_c3zeronet --workload c3 --variant zeronet --steps 2 --warmup 1
  PASSES="trace" run r04_c3dense --workload c3 --variant dense --steps 2 --warmup 1
elif [ "$part" = b ]; then
  PASSES="fetch mfma" run r04_c5 --workload c5 --steps 5 --warmup 2
  PASSES="fetch mfma" run r04_c2 --workload c2 --steps 20 --warmup 3
  PASSES="fetch mfma" run r04_c1 --workload c1 --steps 10 --warmup 2
  PASSES="fetch mfma" run r04_c4 --workload c4 --steps 3 --warmup 1
else
  for w in c3 c2 c1 c4 c5; do
    timeout -k 10 600 python bench.py --workload $w > gpurun_out/r04_bench_$w.json 2> gpurun_out/r04_bench_$w.err || echo "bench $w failed"
    tail -c 160 gpurun_out/r04_bench_$w.json; echo
  done
  for v in pairs zeronet dense; do
    timeout -k 10 600 python bench.py --workload c3 --variant $v > gpurun_out/r04_bench_c3_$v.json 2> gpurun_out/r04_bench_c3_$v.err || echo "bench $v failed"
  done
  timeout -k 10 300 python tools/next_rows_bench.py > gpurun_out/r04_next_rows.jsonl 2> gpurun_out/r04_next_rows.err; tail -2 gpurun_out/r04_next_rows.jsonl
  timeout -k 10 300 python tools/stream_kernels_bench.py > gpurun_out/r04_stream_kernels.jsonl 2> gpurun_out/r04_stream_kernels.err; tail -2 gpurun_out/r04_stream_kernels.jsonl
  bash tools/small_ablate.sh > gpurun_out/r04_small_ablate_run.jsonl 2>&1; tail -2 gpurun_out/r04_small_ablate_run.jsonl
fi
