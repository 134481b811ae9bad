#!/bin/sh
# Per-kernel averages over whole games (moves 6-126) for two builds of the library on one box:
# sh tools/ab_lib_kstats.sh <name of the other build (libazamd_<name>.so)>
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd /tmp && export TMPDIR=/tmp
for lib in libazamd_$1.so libazamd.so; do
  echo "== $lib"
  export AZAMD_LIB=$R/alphazero_amd/$lib
  rm -rf /tmp/abk
  timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abk -- python3 $R/tools/move_time.py 512 64 4 192 48 1 > /tmp/abk.out 2>/tmp/abk.err || { tail -20 /tmp/abk.err; exit 1; }
  tail -1 /tmp/abk.out
  python3 - $(find /tmp/abk -name "*kernel_stats.csv" | head -n 1) <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:9]:
    print(f"{r['Name'].split('(')[0][:60]:62s} n={r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us  max {float(r['MaxNs'])/1e3:8.1f}  {float(r['Percentage']):5.1f}%")
PY
done
