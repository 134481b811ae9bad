#!/bin/sh
# The measurement set behind the figures of a round's final tree: sh tools/final_profiles.sh <tag> <part>   (GPU box, repo root)
#   part a: rocprofv3 passes of the headline (tools/profile_run.sh), the default bench line, the 330-step window, one lane
#   part b: the other BASELINE configurations, few-game moves, one search, kernel stats of configs[1] in one lane
tag=${1:-final}; part=${2:-a}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
O=$R/gpurun_out; mkdir -p $O
line() { python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
print(sys.argv[1].split('/')[-1], '|', d['value'], d['unit'], '|', d['ms_per_step'], 'ms/step | tower', r['achieved'], 'TFLOP/s frac', r['frac'], 'share', r['tower_share_of_wall'], '|', d.get('without_launch_timing'), '|', d.get('games_per_hour_counted'))" $1; }
if [ "$part" = a ]; then
  sh $R/tools/profile_run.sh $tag > $O/${tag}_profile_run.log 2>&1 || { tail -5 $O/${tag}_profile_run.log; exit 1; }
  timeout -k 10 600 python3 $R/bench.py > $O/${tag}_bench_line.json 2>$O/${tag}_bench.err || exit 1
  line $O/${tag}_bench_line.json
  timeout -k 10 300 python3 $R/bench.py --no-cpu-baseline --no-side-legs --steps 330 > $O/${tag}_bench_long_window.json 2>/dev/null || exit 1
  line $O/${tag}_bench_long_window.json
  timeout -k 10 300 python3 $R/bench.py --no-cpu-baseline --no-side-legs --lanes 1 --steps 20 --warmup 4 > $O/${tag}_bench_line_1lane.json 2>/dev/null || exit 1
  line $O/${tag}_bench_line_1lane.json
else
  timeout -k 10 300 python3 $R/bench.py --config 2 --no-cpu-baseline --no-train-step --steps 40 > $O/${tag}_bench_line_config2.json 2>/dev/null || exit 1
  line $O/${tag}_bench_line_config2.json
  timeout -k 10 300 python3 $R/bench.py --config 2 --lanes 1 --no-cpu-baseline --no-train-step --steps 40 > $O/${tag}_bench_line_config2_1lane.json 2>/dev/null || exit 1
  line $O/${tag}_bench_line_config2_1lane.json
  timeout -k 10 600 python3 $R/bench.py --config 5 --no-cpu-baseline --no-train-step > $O/${tag}_bench_line_config5.json 2>/dev/null || exit 1
  line $O/${tag}_bench_line_config5.json
  ( for g in 64 256 512; do
      timeout -k 10 200 python3 $R/bench.py --games $g --lanes 1 --no-cpu-baseline --no-side-legs --steps 30 --warmup 4 > /tmp/fp.json 2>/dev/null || exit 1
      python3 -c "import json; d=json.loads(open('/tmp/fp.json').read().strip().splitlines()[-1]); print('headline network, $g games, one lane:', d['ms_per_step'], 'ms per move,', d['value'], 'sims/s')"
    done ) > $O/${tag}_midsize_games.txt || exit 1
  cat $O/${tag}_midsize_games.txt
  timeout -k 10 200 python3 $R/tools/search_latency.py > $O/${tag}_search_latency.txt 2>/dev/null || exit 1
  tail -4 $O/${tag}_search_latency.txt
  cd /tmp && export TMPDIR=/tmp
  for g in 512 256; do
    rm -rf /tmp/fpk
    timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fpk -- python3 $R/bench.py --config 2 --games $g --lanes 1 --no-cpu-baseline --no-side-legs --steps 20 --warmup 3 > /dev/null 2>/tmp/fpk.err || { tail -5 /tmp/fpk.err; exit 1; }
    cp $(find /tmp/fpk -name "*kernel_stats.csv" | head -n 1) $O/${tag}_config2_${g}games_1lane_kernel_stats.csv
    python3 - $O/${tag}_config2_${g}games_1lane_kernel_stats.csv $g <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].split("(")[0]
    if any(k in n for k in ("k_expand_select", "k_leaf", "k_select", "k_tails", "k_move_tail", "k_tower")):
        print(sys.argv[2], "games:", n[:64], "avg %.1f us" % (float(r["AverageNs"]) / 1e3))
PY
  done
fi
echo final_profiles $part done
