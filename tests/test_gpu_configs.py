"""The BASELINE configurations at (or near) their stated frame counts, on the GPU box.

config 1: the 1000-frame 1280x800 stream, SHA-256 against the manifest made by the REAL
          reference (tests/golden/make_golden_long.py).
config 2: a 2000-frame 1920x1080 16-bit stream through Encoder -> StreamingDecoder with bounded
          memory: decoded == input, and the frame index holds 64-bit offsets beyond 4 GiB.
config 4: a 2000-frame 1280x800 archive, RandomAccessDecoder::DecodeFrames with a shuffled
          index list.
The native driver tests/configs_driver.cc (build/bin/configs_driver) runs the sessions; the
slot ring is reused hundreds of times in each."""
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
    p = subprocess.Popen([DRIVER, "encode", plasma12, "1280", "800", "4", "64", "1000", str(_threads())],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    h, size = hashlib.sha256(), 0
    while True:
        b = p.stdout.read(1 << 22)
        if not b:
            break
        h.update(b)
        size += len(b)
    err = p.stderr.read().decode()
    assert p.wait(timeout=120) == 0, err
    assert size == man["fpv_size"]
    assert h.hexdigest() == man["fpv_sha256"]


def test_config2_long_stream_bounded_memory(synth):
    """2000 frames 1920x1080 16-bit (about 6 GB of stream): Encoder -> StreamingDecoder in 64 MB
    pieces, decoded == input, frame-index offsets (u64, cc:1185-1197) checked past 2^32."""
    with tempfile.TemporaryDirectory(dir="/tmp") as d:
        p = os.path.join(d, "plasma16.raw")
        synth.plasma_frames(32, 1920, 1080, bits=16, seed=9).tofile(p)
        r = subprocess.run([DRIVER, "stream", p, "1920", "1080", "0", "32", "2000", str(_threads())],
                           capture_output=True, text=True, timeout=280)
    assert r.returncode == 0 and r.stdout.startswith("ok stream 2000 frames"), r.stdout + r.stderr
    assert "(> 2^32)" in r.stdout, r.stdout
    print(r.stdout.strip())


def test_config4_archive_shuffled_random_access(plasma12):
    """2000-frame archive, DecodeFrames over a shuffled index list (cc:994-1036), every pixel."""
    r = subprocess.run([DRIVER, "archive", plasma12, "1280", "800", "4", "64", "2000", str(_threads())],
                       capture_output=True, text=True, timeout=280)
    assert r.returncode == 0 and r.stdout.startswith("ok archive 2000 frames"), r.stdout + r.stderr
    print(r.stdout.strip())
