#!/bin/sh
# Few lockstep games of the headline network in one lane against two: sh tools/ab_small_lanes.sh
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
for g in 64 128 256; do for l in 1 2; do
  timeout -k 10 200 python3 $R/bench.py --games $g --lanes $l --no-cpu-baseline --no-side-legs --steps 30 --warmup 4 > /tmp/sl.json 2>/dev/null || exit 1
  python3 -c "import json; d=json.loads(open('/tmp/sl.json').read().strip().splitlines()[-1]); print('games $g lanes $l:', d['ms_per_step'], 'ms/move', d['value'], 'sims/s')"
done; done
