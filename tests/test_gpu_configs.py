"""The BASELINE configurations AT their stated frame counts, on the GPU box.

config 1: the 1000-frame 1280x800 stream, SHA-256 against the manifest made by the REAL
          reference (tests/golden/make_golden_long.py).
config 2: the 10 000-frame 1920x1080 16-bit stream through Encoder -> StreamingDecoder with bounded
          memory: every decoded frame == input, the frame index (64-bit offsets, 29.7 GB of
          stream) checked entry by entry, its last offset asserted; and the whole stream's SHA-256
          against the REAL reference's (make_golden_long.py config2).
config 3: 1000 frames 4096x4096 16-bit through the Encoder over two contexts (FPV_DEVICES=0,0:
          the batch round-robin of the multi-GPU path on the one GPU of the box), SHA-256 against
          the manifest made by the REAL reference (make_golden_long.py config3).
config 4: a 10 000-frame 1280x800 archive, RandomAccessDecoder::DecodeFrames over ALL of it with a
          shuffled index list, sharded over two contexts (FPV_DEVICES=0,0), every pixel compared.
The native driver tests/configs_driver.cc (build/bin/configs_driver) runs the sessions; its "ok"
lines are appended to gpurun_out/configs_full_counts.log (copied to profiles/ per round)."""
import hashlib
import json
import os
import subprocess
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "build", "bin", "configs_driver")


def _log(line):
    d = os.path.join(ROOT, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "configs_full_counts.log"), "a") as f:
        f.write(line.rstrip() + "\n")


def _sha_of_encode(args, env=None, timeout=280):
    """Runs the driver's encode mode and hashes its stdout; returns (sha256, size, seconds)."""
    import time
    t0 = time.time()
    p = subprocess.Popen([DRIVER, "encode"] + [str(a) for a in args], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         env=dict(os.environ, **(env or {})))
    h, size = hashlib.sha256(), 0
    while True:
        b = p.stdout.read(1 << 24)
        if not b:
            break
        h.update(b)
        size += len(b)
    err = p.stderr.read().decode()
    assert p.wait(timeout=timeout) == 0, err
    return h.hexdigest(), size, time.time() - t0


def _threads():
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        return max(4, int(int(q) / int(per))) if q != "max" else 16
    except Exception:
        return 16


@pytest.fixture(scope="module")
def plasma12(synth):
    with tempfile.TemporaryDirectory(dir="/tmp") as d:
        p = os.path.join(d, "plasma12.raw")
        synth.plasma_frames(64, 1280, 800, bits=12, seed=7).tofile(p)
        yield p


def test_config1_1000_frame_stream_sha256(plasma12):
    """cc:1159-1197: 1000 frames through the Encoder; every byte as the reference wrote it."""
    man = json.load(open(os.path.join(ROOT, "tests", "golden", "manifest.json")))["workloads"]["plasma12_1280x800x1000_cycled64_seed7"]
    sha, size, dt = _sha_of_encode([plasma12, 1280, 800, 4, 64, 1000, _threads()])
    assert size == man["fpv_size"]
    assert sha == man["fpv_sha256"]
    _log(f"ok config1 1000 frames 1280x800: {size} bytes, sha256 {sha} == reference, {dt:.1f} s")


def test_config2_10000_frame_stream_bounded_memory(synth):
    """BASELINE config 2 at its count: 10 000 frames 1920x1080 16-bit (29.7 GB of stream): Encoder ->
    StreamingDecoder in 64 MB pieces, every decoded frame == input, every frame-index offset (u64,
    cc:1185-1197) checked against where the chunk went, the last one asserted."""
    with tempfile.TemporaryDirectory(dir="/tmp") as d:
        p = os.path.join(d, "plasma16.raw")
        synth.plasma_frames(32, 1920, 1080, bits=16, seed=9).tofile(p)
        r = subprocess.run([DRIVER, "stream", p, "1920", "1080", "0", "32", "10000", str(_threads())],
                           capture_output=True, text=True, timeout=280)
    assert r.returncode == 0 and r.stdout.startswith("ok stream 10000 frames"), r.stdout + r.stderr
    assert "last index offset 29712630910 (> 2^32)" in r.stdout, r.stdout
    _log("config2 " + r.stdout.strip())


def test_config2_10000_frame_stream_sha256(synth):
    """BASELINE config 2 at its count, byte for byte: the 29.7 GB stream of 10 000 frames 1920x1080
    16-bit (its frame index holds 64-bit offsets beyond 2^32, cc:1185-1197) hashed as it leaves the
    Encoder, against the SHA-256 of what the REAL reference's encode.cc wrote for the same frames
    in the build container (tests/golden/make_golden_long.py config2)."""
    man = json.load(open(os.path.join(ROOT, "tests", "golden", "manifest.json")))["workloads"].get("plasma16_1920x1080x10000_cycled32_seed9")
    assert man, "manifest entry missing: run tests/golden/make_golden_long.py config2 in the build container"
    with tempfile.TemporaryDirectory(dir="/tmp") as d:
        p = os.path.join(d, "plasma16.raw")
        synth.plasma_frames(32, 1920, 1080, bits=16, seed=9).tofile(p)
        sha, size, dt = _sha_of_encode([p, 1920, 1080, 0, 32, 10000, _threads()])
    assert size == man["fpv_size"]
    assert sha == man["fpv_sha256"]
    _log(f"ok config2 10000 frames 1920x1080: {size} bytes, sha256 {sha} == reference, {dt:.1f} s")


def test_config3_1000_frames_4096x4096_two_contexts_sha256(synth):
    """BASELINE config 3 at its count: 1000 frames 4096x4096 16-bit through the Encoder, batches
    alternating over two contexts (FPV_DEVICES=0,0; cc:1159-1197), every byte as the reference
    wrote it (SHA-256 of 18 GB of stream, manifest made by the real reference in the build
    container)."""
    man = json.load(open(os.path.join(ROOT, "tests", "golden", "manifest.json")))["workloads"]["plasma16_4096x4096x1000_cycled8_seed11"]
    with tempfile.TemporaryDirectory(dir="/tmp") as d:
        p = os.path.join(d, "plasma4k.raw")
        synth.plasma_frames(8, 4096, 4096, bits=16, seed=11).tofile(p)
        sha, size, dt = _sha_of_encode([p, 4096, 4096, 0, 8, 1000, _threads()], env={"FPV_DEVICES": "0,0"})
    assert size == man["fpv_size"]
    assert sha == man["fpv_sha256"]
    _log(f"ok config3 1000 frames 4096x4096, FPV_DEVICES=0,0: {size} bytes, sha256 {sha} == reference, {dt:.1f} s, {1000 / dt:.0f} fps incl. hashing")


def test_config4_10000_frame_archive_shuffled_random_access(plasma12):
    """BASELINE config 4 at its count: a 10 000-frame archive, DecodeFrames over a shuffled list of
    ALL its frames (cc:994-1036), sharded over two contexts (FPV_DEVICES=0,0), every pixel."""
    r = subprocess.run([DRIVER, "archive", plasma12, "1280", "800", "4", "64", "10000", str(_threads())],
                       capture_output=True, text=True, timeout=280, env=dict(os.environ, FPV_DEVICES="0,0"))
    assert r.returncode == 0 and r.stdout.startswith("ok archive 10000 frames"), r.stdout + r.stderr
    _log("config4 FPV_DEVICES=0,0 " + r.stdout.strip())
