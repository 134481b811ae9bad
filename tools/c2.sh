for rep in 1 2; do
for L in "" alphazero_amd/libazamd_old.so; do
 echo "[$L] $(AZAMD_LIB=$L timeout -k 10 200 python bench.py --config 2 --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['tower_share_of_wall'])")"
done; done
