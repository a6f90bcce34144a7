#!/bin/bash
# absolute timeline of a cold default run of the upstream benchmark over the library
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
for t in 32 8; do
  for rep in 1 2; do
  echo "== threads $t run $rep"
  FPV_TIMING=1 ./build/bin/ref_benchmark $W/plasma100.raw 1280 800 0 4 100 $t 2>&1 | python3 -u -c "
import sys, time
t0 = time.time()
for line in sys.stdin:
    print('%7.1f ms  %s' % ((time.time() - t0) * 1e3, line.rstrip()[:170]))
"
  done
done
./build/bin/cold_probe 2>&1 | tail -3
