#!/bin/sh
# BASELINE configs[1] (9x9 64ch x 4, 192 sims, 512 games): per-kernel averages under rocprofv3 (one lane), then the bench rate
# with 1 / 2 / 4 lanes, five runs of the default.  sh tools/c2_probe.sh <tag>   (on the GPU box, from the repo root)
tag=${1:-c2}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
O=$R/gpurun_out; mkdir -p $O
rm -rf /tmp/c2p
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c2p -- python3 $R/bench.py --config 2 --lanes 1 --no-cpu-baseline --no-side-legs --steps 10 --warmup 3 > $O/${tag}_line_1lane_under_rocprof.json 2>/tmp/c2p.err || { tail -20 /tmp/c2p.err; exit 1; }
cp $(find /tmp/c2p -name "*kernel_stats.csv" | head -n 1) $O/${tag}_kernel_stats.csv
python3 - $O/${tag}_kernel_stats.csv <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:16]:
    print(f"{r['Name'].split('(')[0][:60]:62s} n={r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us  {float(r['Percentage']):5.1f}%")
PY
for l in 1 2 2 2 2 2 4; do
  timeout -k 10 100 python3 $R/bench.py --config 2 --lanes $l --no-cpu-baseline --no-side-legs --steps 40 > /tmp/c2p_line.json 2>/tmp/c2p_line.err || { tail -20 /tmp/c2p_line.err; exit 1; }
  python3 -c "import json; d=json.loads(open('/tmp/c2p_line.json').read().strip().splitlines()[-1]); print('lanes $l:', d['value'], 'sims/s', d['ms_per_step'], 'ms/move, tower share', d['roofline']['tower_share_of_wall'], 'host', d['host_seconds'])"
  cp /tmp/c2p_line.json $O/${tag}_line_lanes$l.json
done
