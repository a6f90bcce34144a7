#!/bin/bash
# Upstream decode.cc (1 MiB blocks into StreamingDecoder::Decode -- about one frame per call) over this library and over the
# CPU reference, same 300-frame 1280x800 stream (experiments): wall time of the whole process, best of three.
set -u
cd "$(dirname "$0")/../.."
W=$(mktemp -d /tmp/fpv_dec.XXXX); trap 'rm -rf "$W"' EXIT
python - "$W" <<'PY'
import sys, importlib
sys.path.insert(0, '.')
from __graft_entry__ import load_package
load_package(); synth = importlib.import_module('fpv_amd.synth')
fr = synth.plasma_frames(60, 1280, 800, bits=12, seed=7)
with open(sys.argv[1] + "/in.raw", "wb") as f:
    for k in range(300): fr[k % 60].tofile(f)
PY
./build/bin/ref_encode 1280 800 0 4 16 < $W/in.raw > $W/s.fpv || exit 1
ls -la $W/s.fpv | awk '{print "stream bytes", $5}'
for bin in build/bin/ref_decode oracle/_ref/decode build/bin/fpv_decode; do
  [ -x $bin ] || { echo "$bin missing"; continue; }
  best=999999
  for i in 1 2 3; do
    s=$(date +%s%N)
    if [ $bin = build/bin/fpv_decode ]; then $bin 0 4 < $W/s.fpv > $W/out.raw; else $bin 1280 800 0 4 < $W/s.fpv > $W/out.raw; fi
    e=$(date +%s%N); ms=$(( (e - s) / 1000000 )); [ $ms -lt $best ] && best=$ms
  done
  cmp -s $W/out.raw $W/in.raw && ok=identical || ok=DIFFERENT
  echo "$bin: best of 3 ${best} ms for 300 frames = $(( 300000 / best )) fps, output $ok"
done
