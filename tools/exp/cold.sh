python - <<'PY'
import sys, importlib
sys.path.insert(0,'.')
from __graft_entry__ import load_package
load_package(); synth=importlib.import_module('fpv_amd.synth')
synth.plasma_frames(100,1280,800,bits=12).tofile('/tmp/p100.raw')
PY
for t in 8 64; do FPV_TIMING=1 ./build/bin/fpv_benchmark /tmp/p100.raw 1280 800 0 4 100 $t 2>&1 | grep -E "Init|encode"; done
