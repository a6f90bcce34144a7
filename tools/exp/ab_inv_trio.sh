#!/bin/bash
# trio kernel A/B across libfpvhip builds (experiments)
cd "$(dirname "$0")/../.."
export FPV_INVERSE=trio
for lib in "$@"; do
  if [ "$lib" = default ]; then unset FPV_HIP_LIB; else export FPV_HIP_LIB=$PWD/$lib; fi
  echo "== $lib"
  timeout -k 5 120 python tools/exp/inv_bench2.py 1 512 768 1000 1280 2048 2>/dev/null | grep plasma
done
