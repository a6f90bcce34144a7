#!/bin/bash
# A/B of libfpvhip builds on the inverse benchmark (experiments): ab_inv.sh lib1.so lib2.so ... ("default" = in-tree)
cd "$(dirname "$0")/../.."
for rep in 1 2; do
  for lib in "$@"; do
    if [ "$lib" = default ]; then unset FPV_HIP_LIB; else export FPV_HIP_LIB=$PWD/$lib; fi
    echo "== $lib"
    timeout -k 5 120 python tools/exp/inv_bench2.py 1 1000 4096 2>/dev/null | grep plasma
  done
done
