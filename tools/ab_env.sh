#!/bin/sh
# A/B of one build under two settings of an environment variable: tools/ab_env.sh VAR=value [rows...]
kv=$1; shift
for rows in "$@"; do
  for rep in 1 2 3; do
    echo "default: $(timeout -k 10 120 python tools/bench_net.py $rows 192 12 9 2>/dev/null | tail -1)"
    echo "$kv: $(env $kv timeout -k 10 120 python tools/bench_net.py $rows 192 12 9 2>/dev/null | tail -1)"
  done
done
