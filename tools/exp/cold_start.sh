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
for t in 32 64; do echo "== threads $t"; FPV_TIMING=1 timeout 90 ./build/bin/fpv_benchmark $W/plasma100.raw 1280 800 0 4 100 $t 2>&1 | grep -E "fpv-mi355x|encode" | head -14; done
for t in 8 32 64; do timeout 90 ./build/bin/ref_benchmark $W/plasma100.raw 1280 800 0 4 100 $t 2>&1 | grep -E "^total" | sed "s/^/ours threads=$t /"; timeout 90 ./oracle/_ref/benchmark $W/plasma100.raw 1280 800 0 4 100 $t 2>&1 | grep -E "^total" | sed "s/^/ref  threads=$t /"; done
