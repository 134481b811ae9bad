# lockstep self-play: python loop / native loop / native loop + graph replay (launch timing off: e2e.py does not profile)
for rep in 1 2 3; do for m in "AZ_NATIVE_ROUNDS=0" "AZ_GRAPH_ROUNDS=0" "AZ_GRAPH_ROUNDS=1"; do
echo "$m: $(env $m timeout -k 10 200 python tools/e2e.py 512 20 1 16 9 64 4 192 48 2 2>/dev/null | tail -1 | cut -c80-130) | $(env $m timeout -k 10 200 python tools/e2e.py 40 10 1 8 9 192 12 256 48 1 2>/dev/null | tail -1 | cut -c80-130)"
done; done
