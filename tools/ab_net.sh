#!/bin/sh
# A/B of two builds of the library on one box: tools/ab_net.sh <other.so> [rows...]; alternates runs of bench_net.py
other=$1; shift
for rows in "$@"; do
  for rep in 1 2 3; do
    echo "default: $(timeout -k 10 120 python tools/bench_net.py $rows 192 12 9 2>/dev/null | tail -1)"
    echo "other:   $(AZAMD_LIB=$other timeout -k 10 120 python tools/bench_net.py $rows 192 12 9 2>/dev/null | tail -1)"
  done
done
