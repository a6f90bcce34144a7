#!/bin/bash
# End-to-end runs of the BASELINE configurations on the GPU box (bounded samples of
# the frame counts; same geometry, bit depth and content), with the encoder's stage
# statistics (FPV_STATS=1) and the round trip through all three decoders.
# Scratch files go to /tmp and are removed.  usage: tools/configs_e2e.sh [threads]
set -u
cd "$(dirname "$0")/.."
TH=${1:-16}
W=$(mktemp -d /tmp/fpv_cfg.XXXX)
trap 'rm -rf "$W"' EXIT
step() { echo; echo "[$(date +%T)] $*"; }
gen() {  # name xs ys bits distinct total [noise]
python - "$W" "$@" <<'PY'
import sys, importlib
sys.path.insert(0, '.')
from __graft_entry__ import load_package
load_package()
synth = importlib.import_module('fpv_amd.synth')
out, name, xs, ys, bits, distinct, total = sys.argv[1], sys.argv[2], *map(int, sys.argv[3:8])
fr = synth.noise_frames(distinct, xs, ys, seed=1) if len(sys.argv) > 8 else synth.plasma_frames(distinct, xs, ys, bits=bits, seed=7)
with open(f"{out}/{name}.raw", "wb") as f:
    for k in range(total):
        fr[k % distinct].tofile(f)
print(f"{name}: {total} frames {xs}x{ys} ({distinct} distinct), {total * xs * ys * 2 / 1e6:.0f} MB")
PY
}
echo "host: $(nproc) cpus visible, cpu.max = $(cat /sys/fs/cgroup/cpu.max 2>/dev/null), threads = $TH"
export FPV_STATS=1
step "config 0: 256x256 16-bit noise, 64 frames"
gen noise 256 256 16 64 64 noise
timeout -k 10 120 ./build/bin/fpv_benchmark $W/noise.raw 256 256 0 0 64 $TH 2>&1
rm -f $W/noise.raw
step "config 1 / 4: 1280x800 12-bit-in-16 plasma, 1024 frames (encode; random-access + streaming decode of the archive)"
gen plasma12 1280 800 12 64 1024
timeout -k 10 300 ./build/bin/fpv_benchmark $W/plasma12.raw 1280 800 0 4 1024 $TH 2>&1
rm -f $W/plasma12.raw
step "config 2: 1920x1080 16-bit plasma, 512-frame sample of the stream"
gen plasma16 1920 1080 16 32 512
timeout -k 10 300 ./build/bin/fpv_benchmark $W/plasma16.raw 1920 1080 0 0 512 $TH 2>&1
rm -f $W/plasma16.raw
step "config 3: 4096x4096 16-bit plasma, 48-frame sample"
gen plasma4k 4096 4096 16 8 48
timeout -k 10 400 ./build/bin/fpv_benchmark $W/plasma4k.raw 4096 4096 0 0 48 $TH 2>&1
rm -f $W/plasma4k.raw
step done
