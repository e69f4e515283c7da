#!/bin/bash
# GPU box: build and run tools/small_ablate.hip once per ablation (lines of JSON on stdout)
for a in 0 1 2 3; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -DAGGF_SMALL_ABL=$a tools/small_ablate.hip aggforce_amd/csrc/aggf_util.hip -o /tmp/small_ablate_$a -ldl 2>&1 | grep " error"
  timeout -k 10 120 /tmp/small_ablate_$a
done
