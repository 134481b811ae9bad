#!/bin/sh
# same-box A/B of the default library against alphazero_amd/libazamd_<name>.so: tools/ab_lib.sh <name>
L=alphazero_amd/libazamd_$1.so
for i in 1 2 3; do
  echo "default: $(timeout -k 10 120 python tools/bench_net.py 4096 192 12 9 2>/dev/null | tail -1)"
  echo "$1: $(AZAMD_LIB=$L timeout -k 10 120 python tools/bench_net.py 4096 192 12 9 2>/dev/null | tail -1)"
done
sh tools/ab_bench.sh "" "AZAMD_LIB=$L"
