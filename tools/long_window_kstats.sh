#!/bin/sh
# Kernel averages of the headline workload in one lane over a 330-step window (whole games, slots refilling) against the
# first moves: sh tools/long_window_kstats.sh <tag>
tag=${1:-lw}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for steps in 20 330; do
  rm -rf /tmp/lwk
  timeout -s KILL 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lwk -- python3 $R/bench.py --lanes 1 --no-cpu-baseline --no-side-legs --steps $steps --warmup 4 > /tmp/lwk.json 2>/tmp/lwk.err || { tail -5 /tmp/lwk.err; exit 1; }
  cp $(find /tmp/lwk -name "*kernel_stats.csv" | head -n 1) $O/${tag}_1lane_${steps}steps_kernel_stats.csv
  python3 - $O/${tag}_1lane_${steps}steps_kernel_stats.csv /tmp/lwk.json $steps <<'PY'
import csv, json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(f"steps {sys.argv[3]}: {d['value']:.0f} sims/s, tower {d['roofline']['achieved']} TFLOP/s, tower share {d['roofline']['tower_share_of_wall']}")
for r in list(csv.DictReader(open(sys.argv[1])))[:9]:
    print(f"   {r['Name'].split('(')[0][:58]:60s} n={r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.1f} us  {float(r['Percentage']):5.2f}%")
PY
done
