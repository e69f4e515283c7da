#!/bin/bash
# runs on the GPU box: round-3 profiles of every workload (kernel trace + PMC passes)
set -o pipefail
bash tools/profile_gpu.sh r03_c3 --workload c3 --steps 3 --warmup 1 > gpurun_out/r03_c3.log 2>&1; tail -1 gpurun_out/r03_c3.log
PASSES="fetch write mfma" bash tools/profile_gpu.sh r03_c3pairs --workload c3 --variant pairs --steps 3 --warmup 1 > gpurun_out/r03_c3pairs.log 2>&1; tail -1 gpurun_out/r03_c3pairs.log
PASSES="fetch mfma" bash tools/profile_gpu.sh r03_c5 --workload c5 --steps 5 --warmup 2 > gpurun_out/r03_c5.log 2>&1; tail -1 gpurun_out/r03_c5.log
PASSES="fetch mfma" bash tools/profile_gpu.sh r03_c2 --workload c2 --steps 20 --warmup 3 > gpurun_out/r03_c2.log 2>&1; tail -1 gpurun_out/r03_c2.log
PASSES="fetch mfma" bash tools/profile_gpu.sh r03_c1 --workload c1 --steps 10 --warmup 2 > gpurun_out/r03_c1.log 2>&1; tail -1 gpurun_out/r03_c1.log
PASSES="fetch mfma" bash tools/profile_gpu.sh r03_c4 --workload c4 --steps 3 --warmup 1 > gpurun_out/r03_c4.log 2>&1; tail -1 gpurun_out/r03_c4.log
PASSES="trace" bash tools/profile_gpu.sh r03_c3zeronet --workload c3 --variant zeronet --steps 2 --warmup 1 > gpurun_out/r03_c3zeronet.log 2>&1; tail -1 gpurun_out/r03_c3zeronet.log
PASSES="trace" bash tools/profile_gpu.sh r03_c3dense --workload c3 --variant dense --steps 2 --warmup 1 > gpurun_out/r03_c3dense.log 2>&1; tail -1 gpurun_out/r03_c3dense.log
timeout -k 10 300 python tools/next_rows_bench.py > gpurun_out/r03_next_rows.jsonl 2> gpurun_out/r03_next_rows.err; tail -4 gpurun_out/r03_next_rows.jsonl
timeout -k 10 300 python tools/stream_kernels_bench.py > gpurun_out/r03_stream_kernels.jsonl 2> gpurun_out/r03_stream_kernels.err; tail -2 gpurun_out/r03_stream_kernels.jsonl
