#!/bin/bash
# TEST SCRIPT.  End-to-end check on the GPU box: upstream CLI sources linked against
# our library vs the CPU reference binaries (oracle/_ref, the checker), same input, SHA-256.
# Every step runs under its own timeout; big scratch files go to /tmp.
set -u
cd "$(dirname "$0")/.."
W=$(mktemp -d /tmp/fpv_e2e.XXXX)
trap 'rm -rf "$W"' EXIT
T="timeout 90"
step() { echo "[$(date +%T)] $*"; }
python - "$W" <<'PY'
import sys, importlib
sys.path.insert(0, '.')
from __graft_entry__ import load_package
load_package()
synth = importlib.import_module('fpv_amd.synth')
out = sys.argv[1]
synth.plasma_frames(100, 1280, 800, bits=12).tofile(out + '/plasma100.raw')
synth.noise_frames(64, 256, 256, seed=1).tofile(out + '/noise64.raw')
PY
step "host: $(nproc) cpus"
for t in 0 8; do
  step "upstream encode.cc over our library, threads=$t"
  $T ./build/bin/ref_encode 1280 800 0 4 $t < $W/plasma100.raw | sha256sum
done
step "our fpv_encode, threads=16"
$T ./build/bin/fpv_encode 1280 800 0 4 16 < $W/plasma100.raw 2>/dev/null | sha256sum
if [ -x oracle/_ref/encode ]; then
  step "CPU reference encode"
  $T ./oracle/_ref/encode 1280 800 0 4 8 < $W/plasma100.raw | sha256sum
fi
$T ./build/bin/ref_encode 1280 800 0 4 8 < $W/plasma100.raw > $W/p.fpv
ls -l $W/p.fpv
step "upstream decode.cc over our library"
$T ./build/bin/ref_decode 1280 800 0 4 < $W/p.fpv 2>/dev/null | cmp - $W/plasma100.raw && echo "  round trip ok"
step "our fpv_decode"
$T ./build/bin/fpv_decode 0 4 < $W/p.fpv 2>/dev/null | cmp - $W/plasma100.raw && echo "  round trip ok"
if [ -x oracle/_ref/decode ]; then
  step "CPU reference decode of our stream"
  $T ./oracle/_ref/decode 1280 800 0 4 < $W/p.fpv 2>/dev/null | cmp - $W/plasma100.raw && echo "  round trip ok"
fi
step "upstream benchmark.cc over our library (config 0: 256x256 noise x64, 4 threads)"
$T ./build/bin/ref_benchmark $W/noise64.raw 256 256 0 0 64 4 2>&1 | grep -E "total|ok|rror"
step "upstream benchmark.cc over our library, plasma 100 frames"
for t in 8 32 64; do $T ./build/bin/ref_benchmark $W/plasma100.raw 1280 800 0 4 100 $t 2>&1 | grep -E "total|rror" | sed "s/^/threads=$t /"; done
if [ -x oracle/_ref/benchmark ]; then
  step "CPU reference benchmark, same host"
  for t in 1 8 32 64; do $T ./oracle/_ref/benchmark $W/plasma100.raw 1280 800 0 4 100 $t 2>&1 | grep -E "total|rror" | sed "s/^/threads=$t /"; done
fi
step "our fpv_benchmark"
for t in 8 64 128; do $T ./build/bin/fpv_benchmark $W/plasma100.raw 1280 800 0 4 100 $t 2>&1 | sed "s/^/threads=$t /"; done
step done
