#!/bin/sh
# Headline network (9x9 192ch x 12) with few lockstep games: the count on the device picks the 96-row low-latency tower
# (AZ_TOWER_SMALL=0 switches that variant off). sh tools/ab_midsize.sh   (on the GPU box, from the repo root)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
for g in 64 256 384 512; do for v in 0 1; do
  AZ_TOWER_SMALL=$v timeout -k 10 200 python3 $R/bench.py --games $g --lanes 1 --no-cpu-baseline --no-side-legs --steps 20 --warmup 3 > /tmp/mid.json 2>/dev/null || exit 1
  python3 -c "import json; d=json.loads(open('/tmp/mid.json').read().strip().splitlines()[-1]); print('games $g small=$v:', d['value'], 'sims/s', d['ms_per_step'], 'ms/move', d['roofline']['rows_per_launch'], 'rows/launch')"
done; done
