for cfg in "FPV_STRIP_MAJOR=0" "FPV_STRIP_MAJOR=1" "FPV_STRIP_MAJOR=1 FPV_WAVES_PER_CU=16" "FPV_STRIP_MAJOR=1 FPV_WAVES_PER_CU=8" "FPV_STRIP_MAJOR=1 FPV_STRIP_ROWS=16" "FPV_STRIP_MAJOR=1 FPV_STRIP_ROWS=80"; do
  echo "== $cfg"; env $cfg timeout 120 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['kernel_ms'], d['roofline']['achieved'])"
done
