#!/bin/sh
# Headline workload: workgroup limit per tower launch against lane count, runs on one box: sh tools/ab_headline_twg.sh
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
run() {
  timeout -k 10 200 python3 $R/bench.py --no-cpu-baseline --no-side-legs --steps 30 --warmup 4 "$@" > /tmp/hl.json 2>/dev/null || exit 1
  python3 -c "import json; d=json.loads(open('/tmp/hl.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$*:', d['value'], 'sims/s', r['achieved'], 'TFLOP/s, tower share', r['tower_share_of_wall'])"
}
run --lanes 2
run --lanes 2 --tower-workgroups 128
run --lanes 2 --tower-workgroups 64
run --lanes 3 --tower-workgroups 128
run --lanes 4 --tower-workgroups 128
run --lanes 4 --tower-workgroups 64
run --lanes 2 --tower-workgroups 224
run --lanes 2 --tower-workgroups 288
run --lanes 2
