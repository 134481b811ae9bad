#!/bin/sh
# per-kernel totals of a short headline bench under two environments: sh tools/ab_kstats.sh "A=1" "B=2"  (GPU box, repo root)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
i=0
for kv in "$@"; do
  i=$((i+1)); rm -rf /tmp/abk_$i
  env $kv timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abk_$i -- python3 $R/bench.py --no-cpu-baseline --no-side-legs --steps 6 --warmup 2 > /tmp/abk_$i.json 2>/dev/null || exit 1
  echo "== $kv: $(python3 -c "import json; d=json.loads(open('/tmp/abk_$i.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['achieved'])")"
  python3 - $(find /tmp/abk_$i -name "*kernel_stats.csv" | head -n 1) <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    print(f"{r['Name'].split('(')[0][:56]:58s} n={r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:9.1f} us total {float(r['TotalDurationNs'])/1e6:9.1f} ms {float(r['Percentage']):5.1f}%")
PY
done
