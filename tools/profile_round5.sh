#!/bin/bash
# Runs ON THE GPU BOX: round-5 profiles of every workload (kernel trace + separate PMC passes) from ONE tree.
#   bash tools/profile_round5.sh a   -> c3 (+ the instruction-mix pass for K1 / K3) and its variants   (profiles/r05_c3*)
#   bash tools/profile_round5.sh b   -> c1, c2, c4, c5                                                   (profiles/r05_c{1,2,4,5}*)
#   bash tools/profile_round5.sh c   -> bench lines with the CPU baseline, stream kernels, next rows, size sweep
#   bash tools/profile_round5.sh p   -> c3 + pairs: the pack pipeline's terms apart (serial / chunked / overlapped traces)
# Afterwards, in the build container:  for t in ...; do python tools/commit_profile.py r05_$t "<bench args>"; done
set -o pipefail
part=${1:-a}
run() { tag=$1; shift; bash tools/profile_gpu.sh $tag "$@" > gpurun_out/$tag.log 2>&1; tail -1 gpurun_out/$tag.log; }
if [ "$part" = a ]; then
  PASSES="fetch write dram mfma insts" run r05_c3 --workload c3 --steps 3 --warmup 1
  PASSES="trace" run r05_c3zeronet --workload c3 --variant zeronet --steps 2 --warmup 1
  PASSES="trace" run r05_c3dense --workload c3 --variant dense --steps 2 --warmup 1
elif [ "$part" = b ]; then
  PASSES="fetch mfma" run r05_c5 --workload c5 --steps 5 --warmup 2
  PASSES="fetch mfma" run r05_c2 --workload c2 --steps 20 --warmup 3
  PASSES="fetch mfma insts" run r05_c1 --workload c1 --steps 10 --warmup 2
  PASSES="fetch mfma" run r05_c4 --workload c4 --steps 3 --warmup 1
elif [ "$part" = p ]; then
  PASSES="fetch mfma" run r05_c3pairs --workload c3 --variant pairs --steps 3 --warmup 1
  AGGF_GRAM_PACK=serial PASSES="trace" run r05_c3pairs_serial --workload c3 --variant pairs --steps 3 --warmup 1
  AGGF_GRAM_PACK=chunked PASSES="trace" run r05_c3pairs_chunked --workload c3 --variant pairs --steps 3 --warmup 1
else
  for w in c3 c2 c1 c4 c5; do
    timeout -k 10 600 python bench.py --workload $w > gpurun_out/r05_bench_$w.json 2> gpurun_out/r05_bench_$w.err || echo "bench $w failed"
    tail -c 160 gpurun_out/r05_bench_$w.json; echo
  done
  for v in pairs zeronet dense; do
    timeout -k 10 600 python bench.py --workload c3 --variant $v > gpurun_out/r05_bench_c3_$v.json 2> gpurun_out/r05_bench_c3_$v.err || echo "bench $v failed"
  done
  timeout -k 10 300 python tools/next_rows_bench.py > gpurun_out/r05_next_rows.jsonl 2> gpurun_out/r05_next_rows.err; tail -2 gpurun_out/r05_next_rows.jsonl
  timeout -k 10 300 python tools/stream_kernels_bench.py > gpurun_out/r05_stream_kernels.jsonl 2> gpurun_out/r05_stream_kernels.err; tail -2 gpurun_out/r05_stream_kernels.jsonl
  timeout -k 10 600 python tools/size_sweep.py 130 144 160 192 224 250 260 300 320 340 360 400 448 470 500 501 640 700 768 1000 1001 1024 2047 2048 4095 > gpurun_out/r05_size_sweep.jsonl 2> gpurun_out/r05_size_sweep.err
  timeout -k 10 300 python tools/size_sweep.py f32 64 144 260 400 500 501 640 1000 1001 1024 2048 >> gpurun_out/r05_size_sweep.jsonl 2>> gpurun_out/r05_size_sweep.err
  timeout -k 10 300 python tools/size_sweep.py f32f64 144 260 320 384 400 470 500 501 640 1001 1024 2048 4096 >> gpurun_out/r05_size_sweep.jsonl 2>> gpurun_out/r05_size_sweep.err
  : > gpurun_out/r05_midsize.jsonl
  for sz in "304 20 2000000" "582 35 1000000" "860 56 600000"; do
    timeout -k 10 200 python tools/midsize_bench.py $sz >> gpurun_out/r05_midsize.jsonl 2>> gpurun_out/r05_midsize.err
  done
  tail -3 gpurun_out/r05_midsize.jsonl
fi
