"""fpvc::Frame public API (Predict / Compress / CompressPredicted / third
constructor / Uncompress / OutputFull / 8-bit constructor) against fixtures
produced by the real reference.  CPU: host code over the oracle shim;
GPU (-m gpu): the product library."""
import ctypes
import os

import numpy as np
import pytest

from frame_api_driver import FrameApi, cases

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = np.load(os.path.join(HERE, "golden", "frame_api.npz"))
SHIM = os.environ.get("FPV_TEST_SHIM") or os.path.join(HERE, "shim", "libfpv_amd_cpushim.so")


def check(api, synth):
    xs, ys, cs = cases(synth)
    for name, img, delta, shift, be in cs:
        assert np.array_equal(api.compress_full(img, delta, xs, ys, shift, be), GOLD[name + "/full"]), name
        if delta is None:
            continue
        got = api.cycle(img, delta, xs, ys, shift, be)
        for k, v in got.items():
            assert np.array_equal(v, GOLD[name + "/" + k]), f"{name}: {k}"
        # parallel=false takes the other Brotli path, same bytes
        assert np.array_equal(api.cycle(img, delta, xs, ys, shift, be, parallel=False)["sizes"], got["sizes"])
    fr = synth.plasma_frames(2, xs, ys, bits=12, seed=3)
    core = api.compress8_core((fr[1] >> 4).astype(np.uint8), (fr[0] >> 4).astype(np.uint8), xs, ys)
    assert np.array_equal(core, GOLD["frame8/core"])


def test_frame_api_host_logic_cpu(fpv, synth):
    if not os.path.exists(SHIM):
        pytest.skip("tests/shim not built")
    check(FrameApi(ctypes.CDLL(SHIM), "fpvc_"), synth)


@pytest.mark.gpu
def test_frame_api_gpu(fpv, synth):
    check(FrameApi(fpv.host_lib(), "fpvc_"), synth)


def test_uncompress_restores_raw_planes(synth, oracle):
    # property on the fixtures themselves: the life cycle is lossless except for the
    # reference's known NO_LOW corner
    xs, ys, cs = cases(synth)
    for name, img, delta, shift, be in cs:
        if delta is None:
            continue
        h, l, _ = oracle.split(img, shift, be)
        assert np.array_equal(GOLD[name + "/rest_high"], h), name
        if shift != 8 and name != "zero_low_vs_delta":
            assert np.array_equal(GOLD[name + "/rest_low"], l), name


def _check_concurrent(api, synth):
    """The fixtures again, from eight threads at once and in a shuffled order: Predict calls of
    different shifts / byte orders / delta frames meet in the combiner (ctypes drops the GIL)."""
    import random
    from concurrent.futures import ThreadPoolExecutor
    xs, ys, cs = cases(synth)
    jobs = [c for c in cs for _ in range(6)]
    random.Random(5).shuffle(jobs)

    def one(job):
        name, img, delta, shift, be = job
        ok = np.array_equal(api.compress_full(img, delta, xs, ys, shift, be), GOLD[name + "/full"])
        if ok and delta is not None:
            got = api.cycle(img, delta, xs, ys, shift, be)
            ok = all(np.array_equal(v, GOLD[name + "/" + k]) for k, v in got.items())
        return name, ok
    with ThreadPoolExecutor(8) as ex:
        bad = [name for name, ok in ex.map(one, jobs) if not ok]
    assert not bad, bad


def test_frame_api_concurrent_cpu(fpv, synth):
    if not os.path.exists(SHIM):
        pytest.skip("tests/shim not built")
    _check_concurrent(FrameApi(ctypes.CDLL(SHIM), "fpvc_"), synth)


@pytest.mark.gpu
def test_frame_api_concurrent_gpu(fpv, synth):
    _check_concurrent(FrameApi(fpv.host_lib(), "fpvc_"), synth)
