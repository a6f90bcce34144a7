#!/bin/bash
# Round 2's ablation of k_stream: diagnostic builds with parts of the step compiled out
# (hipcc ... -DFPV_ABL=<bits> -o build/ab/libfpvhip_abl<v>.so, see fpv_kernels.hip), each timed on
# the streaming kernels (FPV_FORWARD=stream) in one process per variant.  The library under test is
# named on tools/ab_forward.py's command line: the package does not read FPV_HIP_LIB any more.
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/abl
for v in 0 1 2 4 8 16 32 64 6 14 30 127 256 0; do
  lib=build/ab/libfpvhip_abl$v.so
  [ -f "$lib" ] || { echo "abl=$v: $lib not built" | tee -a gpurun_out/abl/abl.log; continue; }
  timeout -k 10 120 python tools/ab_forward.py --kind plasma --rounds 30 "$lib@FPV_FORWARD=stream" 2>&1 | grep "^{" | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('abl=$v', 'k_stream_ms', d['kernel_ms'][0], 'batch_ms_min', d['ms_min'], 'batch_ms_median', d['ms_median'])" | tee -a gpurun_out/abl/abl.log
done
