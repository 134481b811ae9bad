#!/bin/sh
# rocprofv3 passes behind the figures bench.py prints; run on the GPU box from the repo root: sh tools/profile_run.sh <tag>
# Writes gpurun_out/<tag>_*: kernel stats + trace of a short three-lane bench, the same with one lane, and separate PMC
# passes (FETCH_SIZE, WRITE_SIZE; MFMA / wave counters) as the profiler cannot split counters over passes itself.
tag=${1:-prof}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
O=$R/gpurun_out; mkdir -p $O
B="$R/bench.py --no-cpu-baseline --no-side-legs --steps 4 --warmup 1"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_trace -- python3 $B > $O/${tag}_bench_line_under_rocprof.json 2>/tmp/p_trace.err || exit 1
cp $(find /tmp/p_trace -name "*kernel_stats.csv" | head -1) $O/${tag}_bench_kernel_stats.csv
python3 $R/tools/profile_summary.py trace $(find /tmp/p_trace -name "*kernel_trace.csv" | head -1) > $O/${tag}_tower_trace_summary.json
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_trace1 -- python3 $B --lanes 1 > $O/${tag}_bench_line_1lane_under_rocprof.json 2>/tmp/p_trace1.err || exit 1
cp $(find /tmp/p_trace1 -name "*kernel_stats.csv" | head -1) $O/${tag}_bench_1lane_kernel_stats.csv
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/p_fetch -- python3 $B > /dev/null 2>/tmp/p_fetch.err || exit 1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/p_write -- python3 $B > /dev/null 2>/tmp/p_write.err || exit 1
python3 $R/tools/profile_summary.py pmc $(find /tmp/p_fetch -name "*counter_collection.csv" | head -1) $(find /tmp/p_write -name "*counter_collection.csv" | head -1) > $O/${tag}_tower_hbm_traffic.json
echo profile_run done
