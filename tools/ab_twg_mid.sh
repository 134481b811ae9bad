#!/bin/sh
# Two lanes, mid-size game counts: the driver's workgroup limit per tower launch against 1024: sh tools/ab_twg_mid.sh
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
run() {
  timeout -k 10 200 python3 $R/bench.py --no-cpu-baseline --no-side-legs --steps 30 --warmup 4 "$@" > /tmp/hl.json 2>/dev/null || exit 1
  python3 -c "import json; d=json.loads(open('/tmp/hl.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$*:', d['value'], 'sims/s', d['ms_per_step'], 'ms/move', r['achieved'], 'TFLOP/s, tower share', r['tower_share_of_wall'])"
}
for g in 1024 2048 4096; do
  run --games $g --lanes 2 --tower-workgroups 1024
  run --games $g --lanes 2
done
run --config 5 --tower-workgroups 1024
run --config 5
