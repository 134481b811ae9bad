#!/bin/sh
# Search-kernel averages of the headline workload in one lane (4096 games: throughput-bound, four waves' worth of work per SIMD)
# for several builds of the library: LANES=1 sh tools/ab_lib_headline.sh <build> <build> ...   (main = the regular build)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
L=${LANES:-1}
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  lib=libazamd_$v.so; [ "$v" = main ] && lib=libazamd.so
  export AZAMD_LIB=$R/alphazero_amd/$lib
  rm -rf /tmp/abh
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abh -- python3 $R/bench.py --lanes $L --no-cpu-baseline --no-side-legs --steps 6 --warmup 2 > /tmp/abh.json 2>/tmp/abh.err || { tail -20 /tmp/abh.err; exit 1; }
  python3 - $(find /tmp/abh -name "*kernel_stats.csv" | head -n 1) $lib /tmp/abh.json <<'PY'
import csv,sys,json
out=[]
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name'].split('(')[0]
    for k in ("k_expand_select","k_select","k_leaf","k_expand_backup","k_move_tail","k_tails"):
        if "::"+k+"<" in n: out.append(f"{k} {float(r['AverageNs'])/1e3:.1f}")
d=json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
print(sys.argv[2], "|", d['value'], "sims/s, tower share", d['roofline']['tower_share_of_wall'], "|", " | ".join(out))
PY
done
