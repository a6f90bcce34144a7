"""The native columnar batch adapter (include/columnar_batch/*.h):
encoder -> columnar batches -> decoders for FULL / unshifted / MSB8 / PREVIEW images,
the single-image Batch::ExtractImage path, batch recycling, closing timestamps, the
null batch at Close.  tests/columnar_driver.cc checks every pixel; here it is run over
the CPU test shim (no GPU) and over the product library (GPU), and the upstream
columnar tests, compiled unchanged against OUR columnar headers, run on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "build", "bin")


def _run(path, timeout=300):
    if not os.path.exists(path):
        pytest.skip(f"{os.path.relpath(path, ROOT)} not built (make columnar)")
    return subprocess.run([path], capture_output=True, text=True, timeout=timeout)


def test_columnar_driver_over_cpu_shim():
    r = _run(os.path.join(ROOT, "tests", "shim", "columnar_driver_cpushim"))
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("columnar ok") == 5


@pytest.mark.gpu
def test_columnar_driver_on_gpu():
    r = _run(os.path.join(BIN, "columnar_driver"))
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("columnar ok") == 5


@pytest.mark.gpu
def test_upstream_columnar_tests_over_native_adapter():
    """columnar_batch_decoder_test.cc / _encoder_test.cc of the reference, compiled
    unchanged against include/columnar_batch/ (make columnar, in the build container)."""
    r = _run(os.path.join(BIN, "ref_cbtest_native_decoder"))
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Bad Pixel" not in r.stdout
    assert "Closed Encoder - 345678." in r.stdout and "Closed Decoder - 345678." in r.stdout
    assert r.stdout.count("Got the Image") == 3
    r = _run(os.path.join(BIN, "ref_cbtest_native_encoder"))
    assert r.returncode == 0, r.stdout + r.stderr
