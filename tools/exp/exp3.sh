for cfg in "FPV_GROUP=8 FPV_CACHE_DELTA=0" "FPV_GROUP=8 FPV_CACHE_DELTA=1" "FPV_GROUP=4 FPV_CACHE_DELTA=0" "FPV_GROUP=16 FPV_CACHE_DELTA=0" "FPV_GROUP=1 FPV_CACHE_DELTA=0"; do
  echo "== $cfg"; env $cfg timeout 120 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['kernel_ms'], d['roofline']['achieved'])"
done
