#!/bin/sh
# Headline workload, lane counts and tower workgroup limits on one box: sh tools/ab_headline_lanes.sh
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
run() {
  timeout -k 10 200 python3 $R/bench.py --no-cpu-baseline --no-side-legs --steps 20 --warmup 4 "$@" > /tmp/hl.json 2>/dev/null || exit 1
  python3 -c "import json; d=json.loads(open('/tmp/hl.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$*:', d['value'], 'sims/s', r['achieved'], 'TFLOP/s, tower share', r['tower_share_of_wall'], 'rows/launch', r['rows_per_launch'])"
}
run --lanes 2
run --lanes 3
run --lanes 2 --tower-workgroups 512
run --lanes 2 --tower-workgroups 2048
run --lanes 2
run --lanes 4
