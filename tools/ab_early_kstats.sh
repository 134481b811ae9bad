#!/bin/sh
# k_expand_select / k_select / k_leaf averages over the first moves of BASELINE configs[1] for several builds of the library:
# sh tools/ab_early_kstats.sh <build> <build> ...   ("" = the regular build)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  lib=libazamd_$v.so; [ "$v" = main ] && lib=libazamd.so
  export AZAMD_LIB=$R/alphazero_amd/$lib
  rm -rf /tmp/abk
  timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abk -- python3 $R/bench.py --config 2 --lanes 1 --no-cpu-baseline --no-side-legs --steps 20 --warmup 3 > /tmp/abk.out 2>/tmp/abk.err || { tail -20 /tmp/abk.err; exit 1; }
  python3 - $(find /tmp/abk -name "*kernel_stats.csv" | head -n 1) $v <<'PY'
import csv,sys
out=[]
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name'].split('(')[0]
    for k in ("k_expand_select","k_select","k_leaf","k_expand_backup"):
        if "::"+k+"<" in n: out.append(f"{k} {float(r['AverageNs'])/1e3:.1f}")
print(sys.argv[2], "|", " | ".join(out))
PY
done
