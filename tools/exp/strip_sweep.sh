#!/bin/bash
# k_stream tuning sweep (experiments): strip height x persistent waves per CU.
cd "$(dirname "$0")/../.."
for sr in 16 32 80 160 400 800; do
  for w in 8 12 16; do
    echo -n "strip_rows=$sr waves=$w : "
    FPV_STRIP_ROWS=$sr FPV_WAVES_PER_CU=$w timeout -k 5 120 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(r['value'], r['kernel_ms'])"
  done
done
