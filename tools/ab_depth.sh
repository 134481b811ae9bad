for cfg in "4096 128 10 9" "1024 128 10 9" "256 128 10 9" "1 128 10 9" "2048 128 10 19" "512 128 10 19" "256 64 4 9" "1 64 4 9"; do
  for rep in 1 2; do
    echo "depth3: $(timeout -k 10 120 python3 tools/bench_net.py $cfg 2>/dev/null | tail -n 1 | cut -c1-90)"
    echo "depth1: $(AZAMD_LIB=alphazero_amd/libazamd_d1.so timeout -k 10 120 python3 tools/bench_net.py $cfg 2>/dev/null | tail -n 1 | cut -c1-90)"
  done
done
