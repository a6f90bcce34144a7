for cfg in "FPV_TILE_ROWS=16 FPV_XCD_MAP=0" "FPV_TILE_ROWS=20 FPV_XCD_MAP=0" "FPV_TILE_ROWS=20 FPV_XCD_MAP=1" "FPV_TILE_ROWS=8 FPV_XCD_MAP=0" "FPV_TILE_ROWS=32 FPV_XCD_MAP=0" "FPV_TILE_ROWS=4 FPV_XCD_MAP=1" ; do
  echo "== $cfg"; env $cfg timeout 120 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['kernel_ms'])"
done
