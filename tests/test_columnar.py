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


# ---- format parity with the reference's Batch (cb/columnar_batch.cc:31-87) --------------------
#
# tests/golden/columnar.npz holds what the UPSTREAM Batch contains after four predicted frames
# were appended single-threaded (tests/golden/make_golden_columnar.py: upstream columnar_batch.cc
# over the upstream library, dumped by tests/columnar_dump.cc).  The native Batch, fed the same
# frames through the same public calls, must hold the same timestamps, flags, offset arrays and
# stream bytes, in the same region order.  Deliberate differences, checked below as such:
#   * capacity: upstream reserves ONE plane's (and one preview's) worth of stream bytes for the
#     whole batch and stores an empty stream for a frame that no longer fits; the native batch
#     reserves batch_size times that, so the stream columns start further apart;
#   * the schema's compressed delta planes: upstream's are always empty (it sizes the output
#     buffers with reserve(), so CompressPredicted gets zero-byte buffers, cb/columnar_batch.cc:9-24);
#     the native schema carries the real Brotli streams.

def _columnar_parity(dump_binary):
    import sys
    import tempfile

    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from make_golden_columnar import BATCH, N, SHIFT, XS, YS, parse_dump
    if not os.path.exists(dump_binary):
        pytest.skip(f"{os.path.relpath(dump_binary, ROOT)} not built (make columnar)")
    g = np.load(os.path.join(ROOT, "tests", "golden", "columnar.npz"))
    with tempfile.TemporaryDirectory() as d:
        raw = os.path.join(d, "frames.raw")
        g["frames"].tofile(raw)
        r = subprocess.run([dump_binary, raw, str(XS), str(YS), str(SHIFT), str(N), str(BATCH)], capture_output=True, timeout=120)
    assert r.returncode == 0, r.stderr[-500:]
    o = parse_dump(r.stdout)
    assert o["n"] == int(g["n"]) == N and o["batch_size"] == BATCH
    for k in ("timestamps", "flags", "offsets", "preview", "high", "low"):
        assert np.array_equal(o[k], g[k]), k
    assert int(g["offsets"][0][N]) > 0 and (g["offsets"][:, 1:] > g["offsets"][:, :-1]).all()   # every stream fits upstream's capacity
    # layout: upstream regions = timestamps | flags | 3 offset arrays | preview | high | low, each rounded up to 64 bytes
    r64 = lambda v: (v + 63) // 64 * 64
    ref = [int(v) for v in g["regions"]]
    fixed = [0, r64(8 * BATCH), r64(8 * BATCH) + r64(BATCH)]
    fixed += [fixed[2] + r64(4 * (BATCH + 1)), fixed[2] + 2 * r64(4 * (BATCH + 1)), fixed[2] + 3 * r64(4 * (BATCH + 1))]
    assert ref[:6] == fixed                                             # what include/columnar_batch/columnar_batch.h documents
    ours = [int(v) for v in o["regions"]]
    assert ours[5] == 0 and ours[6] > 0 and ours[7] > ours[6]           # same column order: preview, high, low
    assert ours[6] % 64 == 0 and ours[7] % 64 == 0
    # deliberate difference 1: stream capacity per batch (see above)
    assert ours[6] >= BATCH * (ref[6] - ref[5]) - 64 * BATCH and ours[7] - ours[6] >= BATCH * (ref[7] - ref[6]) - 64 * BATCH
    # deliberate difference 2: the schema's delta planes
    assert len(g["delta_high"]) == 0 and len(g["delta_low"]) == 0
    assert len(o["delta_high"]) > 0 and len(o["delta_low"]) > 0
    return o


def test_columnar_format_matches_reference_over_cpu_shim():
    _columnar_parity(os.path.join(ROOT, "tests", "shim", "columnar_dump_cpushim"))


@pytest.mark.gpu
def test_columnar_format_matches_reference_on_gpu():
    _columnar_parity(os.path.join(BIN, "columnar_dump"))


def test_adapter_gpu_failures_are_latched_not_fatal():
    """Frame API, ColumnarBatchEncoder and ColumnarBatchDecoder with injected GPU failures (the CPU
    shim's FPV_SHIM_FAIL_* knobs: a failing fpvhip_predict_submit / _wait / inverse submit): nothing
    aborts; the failing call reports through the channel its signature has (the futures of the
    columnar classes carry an exception; Frame's void methods latch Frame::LastError and leave the
    frame as it was), later calls are refused / no-ops, Close still answers, and after
    Frame::ClearError() -- or with a fresh encoder / decoder object -- the same work gives the
    same bytes (tests/adapter_fail_driver.cc).  Reference convention: cc:160-181."""
    exe = os.path.join(ROOT, "tests", "shim", "adapter_fail_cpushim")
    if not os.path.exists(exe):
        pytest.skip("tests/shim/adapter_fail_cpushim not built (make columnar)")
    env = dict(os.environ)
    env.pop("FPV_ABORT_ON_ERROR", None)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "adapter failures ok" in r.stdout, r.stdout + r.stderr[-3000:]
    for who in ("Frame API failed", "ColumnarBatchEncoder failed", "ColumnarBatchDecoder failed"):
        assert who in r.stderr, who
    # FPV_ABORT_ON_ERROR=1 keeps the old behaviour
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=dict(env, FPV_ABORT_ON_ERROR="1"))
    assert r.returncode != 0 and "fatal" in r.stderr
