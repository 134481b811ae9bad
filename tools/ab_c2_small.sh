#!/bin/sh
# 64-channel tower, 96-row low-latency variant against the 256-row kernel on BASELINE configs[1] (9x9 64ch x 4, 512 games in two
# lanes of 256): sh tools/ab_c2_small.sh   (on the GPU box, from the repo root). Prints the tower's kernel-stats line and the bench rate.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
for v in 0 1 0 1; do
  export AZ_TOWER_SMALL=$v
  rm -rf /tmp/c2s_$v
  timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c2s_$v -- python3 $R/bench.py --config 2 --no-cpu-baseline --no-side-legs --steps 10 --warmup 3 > /tmp/c2s_$v.json 2>/tmp/c2s.err || exit 1
  echo "AZ_TOWER_SMALL=$v (two lanes, traced)"; grep k_tower $(find /tmp/c2s_$v -name "*kernel_stats.csv" | head -n 1) | cut -c1-110
  for l in 2 1; do
    timeout -k 10 100 python3 $R/bench.py --config 2 --lanes $l --no-cpu-baseline --no-side-legs --steps 40 > /tmp/c2s_line.json 2>/dev/null || exit 1
    python3 -c "import json; d=json.loads(open('/tmp/c2s_line.json').read().strip().splitlines()[-1]); print('  lanes $l:', d['value'], 'sims/s', d['ms_per_step'], 'ms/move')"
  done
done
