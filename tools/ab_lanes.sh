#!/bin/sh
# Lane counts against lockstep game counts on the headline network, with and without the 96-row tower (AZ_TOWER_SMALL=0):
# sh tools/ab_lanes.sh   (on the GPU box, from the repo root)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
for gl in "512 2" "768 2" "768 3" "1024 2" "1024 3" "1024 4"; do set -- $gl
 for v in 0 1; do
  AZ_TOWER_SMALL=$v timeout -k 10 200 python3 $R/bench.py --games $1 --lanes $2 --no-cpu-baseline --no-side-legs --steps 20 --warmup 3 > /tmp/l.json 2>/dev/null || exit 1
  python3 -c "import json; d=json.loads(open('/tmp/l.json').read().strip().splitlines()[-1]); print('games $1 lanes $2 small=$v:', d['value'], 'sims/s', d['ms_per_step'], 'ms/move', d['roofline']['rows_per_launch'], 'rows/launch')"
 done
done
