#!/bin/sh
# Two builds of the library on one box: sh tools/ab_lib.sh <name of the other build (libazamd_<name>.so)>
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
for lib in libazamd_$1.so libazamd.so; do
  echo "== $lib"
  export AZAMD_LIB=$R/alphazero_amd/$lib
  timeout -k 10 200 python3 $R/tools/move_time.py 512 64 4 192 48 1 2>/dev/null | tail -1 || exit 1
  timeout -k 10 200 python3 $R/tools/move_time.py 512 64 4 192 48 2 graph 2>/dev/null | tail -1 || exit 1
  timeout -k 10 200 python3 $R/tools/move_time.py 256 192 12 256 48 1 2>/dev/null | tail -1 || exit 1
  timeout -k 10 100 python3 $R/bench.py --config 2 --lanes 2 --no-cpu-baseline --no-side-legs --steps 40 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench config 2, 2 lanes:', d['ms_per_step'], 'ms/move')" || exit 1
  timeout -k 10 100 python3 $R/bench.py --config 2 --lanes 1 --no-cpu-baseline --no-side-legs --steps 40 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench config 2, 1 lane:', d['ms_per_step'], 'ms/move')" || exit 1
done
