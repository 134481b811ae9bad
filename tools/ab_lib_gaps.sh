#!/bin/sh
# Idle gaps between the kernels of whole games for two builds of the library: sh tools/ab_lib_gaps.sh <other build>
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd /tmp && export TMPDIR=/tmp
for lib in libazamd_$1.so libazamd.so; do
  echo "== $lib"
  export AZAMD_LIB=$R/alphazero_amd/$lib
  rm -rf /tmp/abk
  timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/abk -- python3 $R/tools/move_time.py 512 64 4 192 48 1 > /tmp/abk.out 2>/tmp/abk.err || { tail -20 /tmp/abk.err; exit 1; }
  tail -1 /tmp/abk.out
  python3 $R/tools/gap_census.py $(find /tmp/abk -name "*kernel_trace.csv" | head -n 1) 400 || exit 1
done
