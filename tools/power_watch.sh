#!/bin/sh
# Samples the card's socket power and XCD clocks (amd-smi, every ~0.5 s) while a longer bench window runs:
# sh tools/power_watch.sh [steps] [extra bench flags] -> gpurun_out/power_watch.txt, bench figures on stdout.
steps=${1:-150}; shift
O=gpurun_out; mkdir -p $O
( for i in $(seq 1 70); do amd-smi metric -p -c 2>/dev/null | awk '/SOCKET_POWER/{p=$2} /GFX_[0-7]:/{g=1} g&&/ CLK:/{c=c" "$2; g=0} END{print p" W;"c}'; sleep 0.3; done ) > $O/power_watch.txt 2>&1 &
W=$!
timeout -k 10 300 python bench.py --no-cpu-baseline --no-side-legs --steps $steps --warmup 2 "$@" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'])"
kill $W 2>/dev/null
wait $W 2>/dev/null
echo done
