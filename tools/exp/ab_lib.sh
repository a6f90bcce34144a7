#!/bin/bash
# A/B of libfpvhip builds (experiments): tools/exp/ab_lib.sh lib1.so lib2.so ...  ("default" = in-tree library)
cd "$(dirname "$0")/../.."
for rep in 1 2; do
  for lib in "$@"; do
    if [ "$lib" = default ]; then unset FPV_HIP_LIB; else export FPV_HIP_LIB=$PWD/$lib; fi
    echo -n "$lib : "
    timeout -k 5 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(r['value'], r['kernel_ms'], r['roofline']['measured_copy_GBps'])"
  done
done
