#!/bin/sh
# bench.py (short window, no side legs) under several settings: tools/ab_bench.sh "ENV=V --flag x" "..." ...
for rep in 1 2; do
for cfg in "$@"; do
  envs=""; flags=""
  for tok in $cfg; do case $tok in *=*) envs="$envs $tok";; *) flags="$flags $tok";; esac; done
  out=$(env $envs timeout -k 10 200 python bench.py --no-cpu-baseline --no-side-legs --steps 10 --warmup 2 $flags 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['tower_share_of_wall'])")
  echo "[$cfg] $out"
done
done
