#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/abl
for v in 0 1 2 4 8 16 32 64 6 14 30 127 256 0; do
  FPV_HIP_LIB=$PWD/build/ab/libfpvhip_abl$v.so timeout -k 10 120 python tools/ab_forward.py --kind plasma --rounds 30 2>&1 | grep "^{" | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('abl=$v', 'k_stream_ms', d['kernel_ms'][0], 'batch_ms', d['ms_per_batch_best'])" | tee -a gpurun_out/abl/abl.log
done
