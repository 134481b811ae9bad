#!/bin/sh
# quick check of a tower change: bitwise / exact-integer network tests, the 4096-row micro-benchmark, a short bench
timeout -k 10 300 python -u -m pytest tests/test_gpu_network.py -x -q -m gpu -k "exact or bitwise or bf16_path" 2>&1 | tail -3 &&
for i in 1 2; do timeout -k 10 120 python tools/bench_net.py 4096 192 12 9 2>/dev/null | tail -1; done &&
sh tools/ab_bench.sh "" 
