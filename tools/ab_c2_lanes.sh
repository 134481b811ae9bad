#!/bin/sh
# BASELINE configs[1] (512 games of 9x9 64ch x 4, 192 sims) against the lane count, first 126 moves: sh tools/ab_c2_lanes.sh
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
for l in 1 2 3 4; do
  timeout -k 10 200 python3 $R/tools/move_time.py 512 64 4 192 48 $l graph 2>/dev/null | head -1 || exit 1
done
