"""Sanitizer runs of the host codec (CPU only: GPU sanitizers are unavailable on
this pool).  The host sources are compiled together with the oracle-backed shim
of the C ABI, so what runs here is the product's own threading and parsing code."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_threadsanitizer_host_pipeline():
    """Encoder batch pipeline (dispatcher, workers, in-order emitter), batched
    random-access decode and streaming decode under ThreadSanitizer."""
    r = subprocess.run(["make", "-C", ROOT, "tsan"], capture_output=True, text=True, timeout=900)
    if "cannot find -ltsan" in r.stderr or "libtsan" in r.stderr and "No such file" in r.stderr:
        pytest.skip("no libtsan in this toolchain")
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count(" ok, ") == 4 and "concurrent sessions ok" in r.stdout
    assert "concurrent DecodeFrames on one decoder ok" in r.stdout   # used to deadlock (two slots per engine)
    assert "eight devices, one slow: ok" in r.stdout                 # one dispatcher thread per device
    assert "latched GPU failure: ok" in r.stdout                     # Encoder::LastError instead of abort()
    assert r.stdout.count("columnar ok") == 5   # the columnar adapter's queues and futures
    assert "adapter failures ok" in r.stdout    # latched GPU failures in the Frame API / columnar adapters (futures carrying exceptions)
