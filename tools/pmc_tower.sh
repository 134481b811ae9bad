#!/bin/sh
# MFMA / wave counters of un-overlapped 4096-position tower launches, for the default kernel choice and with the
# two-workgroups-per-CU kernel switched off: sh tools/pmc_tower.sh <tag>   (on the GPU box, from the repo root)
tag=${1:-pmc}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
O=$R/gpurun_out; mkdir -p $O
C="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmc_a -- python3 $R/tools/bench_net.py 4096 192 12 9 > /tmp/pmc_a.out 2>&1 || exit 1
python3 $R/tools/pmc_tower.py $(find /tmp/pmc_a -name "*counter_collection.csv" | head -1) > $O/${tag}_tower_pmc_two_per_cu.json
export AZ_TOWER_DUO=0
timeout -s KILL 120 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmc_b -- python3 $R/tools/bench_net.py 4096 192 12 9 > /tmp/pmc_b.out 2>&1 || exit 1
python3 $R/tools/pmc_tower.py $(find /tmp/pmc_b -name "*counter_collection.csv" | head -1) > $O/${tag}_tower_pmc_256_row.json
tail -n 1 /tmp/pmc_a.out; tail -n 1 /tmp/pmc_b.out
