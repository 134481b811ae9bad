#!/bin/sh
# per-kernel average durations of a command under rocprofv3: sh tools/kstats.sh <python script and args>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
rm -rf /tmp/p_ks
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_ks -- python3 $R/"$@" > /dev/null 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('/tmp/p_ks/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(f"{r['Name'].split('(')[0][:48]:50s} n={r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us  {float(r['Percentage']):5.1f}%")
PY
