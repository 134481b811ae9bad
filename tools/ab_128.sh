#!/bin/sh
# 128-channel towers, fused kernels against their alternatives on one box: sh tools/ab_128.sh   (on the GPU box, from the repo root)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
run() { timeout -k 10 60 python3 $R/tools/bench_net.py "$@" | cut -c1-110; }
for r in 4096 1024 768 600; do echo "two-per-CU (above one round of the 256-row kernel): $(run $r 128 10 9)"; echo "256-row kernel only (AZ_TOWER_DUO128=0):            $(AZ_TOWER_DUO128=0 run $r 128 10 9)"; done
for r in 1 256 512; do echo "96-row 256-thread form:                      $(run $r 128 10 9)"; echo "96-row 512-thread form (AZ_TOWER_SMALL_LEAN=0): $(AZ_TOWER_SMALL_LEAN=0 run $r 128 10 9)"; done
for n in 11 13; do echo "fused two-per-CU:              $(run 512 128 10 $n)"; echo "layer path (AZ_NO_FUSED_13=1): $(AZ_NO_FUSED_13=1 run 512 128 10 $n)"; done
for n in 14 15 16; do echo "fused 256-row:                 $(run 512 128 10 $n)"; echo "layer path (AZ_NO_FUSED_16=1): $(AZ_NO_FUSED_16=1 run 512 128 10 $n)"; done
