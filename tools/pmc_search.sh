#!/bin/sh
# Bytes on the L2's memory side per launch of the search / rules / tail kernels (separate FETCH_SIZE and WRITE_SIZE passes over
# the headline workload in one lane, 4096 games): sh tools/pmc_search.sh <tag>   (GPU box, repo root)
tag=${1:-pmc}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
O=$R/gpurun_out; mkdir -p $O
B="$R/bench.py --no-cpu-baseline --no-side-legs --lanes 1 --steps 4 --warmup 1"
rm -rf /tmp/ps_fetch /tmp/ps_write
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/ps_fetch -- python3 $B > /dev/null 2>/tmp/ps_fetch.err || { tail -5 /tmp/ps_fetch.err; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/ps_write -- python3 $B > /dev/null 2>/tmp/ps_write.err || { tail -5 /tmp/ps_write.err; exit 1; }
python3 $R/tools/profile_summary.py pmc_search $(find /tmp/ps_fetch -name "*counter_collection.csv" | head -1) $(find /tmp/ps_write -name "*counter_collection.csv" | head -1) > $O/${tag}_search_kernels_hbm_traffic.json || exit 1
cat $O/${tag}_search_kernels_hbm_traffic.json
