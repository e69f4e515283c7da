#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel trace + separate PMC passes of bench.py.
# Usage: bash tools/profile_gpu.sh <tag> [bench args...]     outputs under gpurun_out/<tag>_*
# PMC passes are separate runs (gpurun refuses --pmc combined with trace domains); PASSES="trace" takes the
# kernel trace only, PASSES="fetch mfma" a subset of the counter passes (default: fetch write dram mfma; "insts" = wave
# cycles, wait / issue-stall / active buckets and the instruction mix of every kernel).
set -o pipefail
tag=$1; shift
bench_args=("$@")
export TMPDIR=/tmp
out=gpurun_out
rm -rf $out/${tag}_trace $out/${tag}_fetch $out/${tag}_write $out/${tag}_dram $out/${tag}_mfma $out/${tag}_insts
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_trace -- python3 bench.py --no-cpu-baseline "${bench_args[@]}" > $out/${tag}_trace.log 2>&1 || { tail -5 $out/${tag}_trace.log; exit 1; }
want="${PASSES:-fetch write dram mfma}"
if [ "$want" != "trace" ]; then
  for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "dram TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum" "mfma SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES" "insts SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM"; do
    read -r name counters <<< "$pass"
    case " $want " in *" $name "*) ;; *) continue ;; esac
    rocprofv3 --pmc $counters --output-format csv -d $out/${tag}_$name -- python3 bench.py --no-cpu-baseline "${bench_args[@]}" --steps 1 --warmup 1 > $out/${tag}_$name.log 2>&1 || { tail -5 $out/${tag}_$name.log; }
  done
fi
python3 tools/summarize_profile.py $tag
tail -2 $out/${tag}_trace.log
