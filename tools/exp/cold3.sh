#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
W=$(mktemp -d /tmp/fpv_cold.XXXX); trap 'rm -rf "$W"' EXIT
python - "$W" <<'PY'
import sys, importlib
sys.path.insert(0, '.')
from __graft_entry__ import load_package
load_package()
synth = importlib.import_module('fpv_amd.synth')
synth.plasma_frames(100, 1280, 800, bits=12).tofile(sys.argv[1] + '/plasma100.raw')
PY
run() { timeout 90 "$@" $W/plasma100.raw 1280 800 0 4 100 64 2>&1 | grep -E "^total" | sed -E 's/.*time: ([0-9.]+) ms.*/\1 ms/'; }
for rep in 1 2; do
echo "ours 64 default:        $(run ./build/bin/ref_benchmark)"
echo "ours 64 mmap_thresh=64M: $(MALLOC_MMAP_THRESHOLD_=67108864 MALLOC_TRIM_THRESHOLD_=268435456 run ./build/bin/ref_benchmark)"
echo "ours 64 FPV_SLOTS=7:     $(FPV_SLOTS=7 run ./build/bin/ref_benchmark)"
echo "ours 64 FPV_BATCH=8:     $(FPV_BATCH=8 run ./build/bin/ref_benchmark)"
echo "ref  64 default:        $(run ./oracle/_ref/benchmark)"
echo "ref  64 mmap_thresh=64M: $(MALLOC_MMAP_THRESHOLD_=67108864 MALLOC_TRIM_THRESHOLD_=268435456 run ./oracle/_ref/benchmark)"
done
FPV_STATS=1 FPV_TIMING=1 timeout 90 ./build/bin/ref_benchmark $W/plasma100.raw 1280 800 0 4 100 64 2>&1 | grep -E "fpv-mi355x: (stats|Init)" | cut -c1-400
