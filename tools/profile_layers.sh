#!/bin/sh
# rocprofv3 passes over the layer-by-layer bf16 network at BASELINE configs[4]'s shape (19x19, 256 channels x 20 blocks,
# 512 boards) and the --config 5 bench line: sh tools/profile_layers.sh <tag>   (on the GPU box, from the repo root)
# Writes gpurun_out/<tag>_layer_path_kernel_stats.csv, <tag>_conv3_duo_pmc.json, <tag>_bench_line_config5.json.
tag=${1:-layers}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
O=$R/gpurun_out; mkdir -p $O
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pl_trace -- python3 $R/tools/bench_net.py 512 256 20 19 > $O/${tag}_layer_path_bench_net.txt 2>/tmp/pl_trace.err || exit 1
cp $(find /tmp/pl_trace -name "*kernel_stats.csv" | head -1) $O/${tag}_layer_path_kernel_stats.csv
C="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pl_pmc -- python3 $R/tools/bench_net.py 512 256 20 19 > /tmp/pl_pmc.out 2>&1 || exit 1
python3 $R/tools/pmc_tower.py $(find /tmp/pl_pmc -name "*counter_collection.csv" | head -1) 1000 k_conv3_duo > $O/${tag}_conv3_duo_pmc.json
cd $R && timeout -k 10 300 python3 bench.py --config 5 --no-cpu-baseline > $O/${tag}_bench_line_config5.json 2>/tmp/pl_bench.err || exit 1
echo profile_layers done
