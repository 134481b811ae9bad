#!/bin/sh
# rocprofv3 passes behind the figures bench.py prints; run on the GPU box from the repo root: sh tools/profile_run.sh <tag>
# Writes gpurun_out/<tag>_*: kernel stats + trace of a short default (two-lane) bench, the same with one lane, and separate PMC
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
python3 $R/tools/profile_summary.py pmc $(find /tmp/p_fetch -name "*counter_collection.csv" | head -1) $(find /tmp/p_write -name "*counter_collection.csv" | head -1) > /tmp/p_traffic.json || exit 1
# the launches use a fixed grid and read the row count on the device: rows per launch come from the line of the traced run
python3 - /tmp/p_traffic.json $O/${tag}_bench_line_under_rocprof.json > $O/${tag}_tower_hbm_traffic.json <<'PY'
import json, sys
t = json.load(open(sys.argv[1]))
line = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
t["rows_per_launch"] = line["roofline"]["rows_per_launch"]
t["kernel"] = line["roofline"]["kernel"]
t["note"] = "bench.py --no-cpu-baseline --no-side-legs --steps 4 --warmup 1; separate FETCH_SIZE and WRITE_SIZE passes; rows_per_launch from the JSON line of the kernel-trace run of the same command"
print(json.dumps(t, indent=1))
PY
echo profile_run done
