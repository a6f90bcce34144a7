"""Pins the CPU oracle (oracle/fpv_oracle.c) to fixtures produced by the real
reference (tests/golden/make_golden.py), and -- where oracle/_ref is present --
to the reference itself on fresh random inputs.  CPU only."""
import hashlib
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = np.load(os.path.join(HERE, "golden", "edge_cases.npz"))
MAN = json.load(open(os.path.join(HERE, "golden", "manifest.json")))
NAMES = [str(n) for n in GOLD["names"]]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize("name", NAMES)
def test_edge_case_planes_flags_and_stream(oracle, name):
    frames, delta = GOLD[name + "/frames"], GOLD[name + "/delta"]
    xs, ys, shift, be = (int(v) for v in GOLD[name + "/meta"])
    for k in range(frames.shape[0]):
        h, l, pv, f = oracle.predict(frames[k], xs, ys, shift, bool(be), delta)
        assert f == GOLD[name + "/flags"][k]
        assert np.array_equal(h, GOLD[name + "/high"][k])
        if shift != 8:
            assert np.array_equal(l, GOLD[name + "/low"][k])
        else:
            assert GOLD[name + "/low"].shape[1] == 0     # reference has no low plane at all
        assert np.array_equal(pv, GOLD[name + "/preview"][k])
    h, _, _, f = oracle.predict(delta, xs, ys, shift, bool(be), None)
    assert f == GOLD[name + "/delta_flags"][0] and np.array_equal(h, GOLD[name + "/delta_high"])
    fpv = oracle.encode_stream(frames, delta, xs, ys, shift, bool(be))
    assert np.array_equal(fpv, GOLD[name + "/fpv"])
    assert np.array_equal(oracle.decode_stream(GOLD[name + "/fpv"]), GOLD[name + "/decoded"])


def test_known_lossy_case_is_reproduced(oracle):
    # all-zero low plane against a delta frame with low bytes: NO_LOW_BYTES was
    # fixed before the delta subtraction, the reference drops the plane.
    name = "zero_low_vs_delta_low"
    assert list(GOLD[name + "/flags"]) == [7, 7]
    assert not np.array_equal(GOLD[name + "/decoded"], GOLD[name + "/frames"])


def test_noise_config0_flags_and_stream(oracle, synth):
    w = MAN["workloads"]["noise_256x256x64_seed1"]
    nz = synth.noise_frames(64, 256, 256, seed=1)
    assert sha(nz) == w["input_sha256"]
    flags = [oracle.predict(nz[k], 256, 256, 0, False, nz[0])[3] for k in range(64)]
    assert flags == w["flags"]
    assert 10 < sum(1 for f in flags if f & 2) < 54      # heuristic flips both ways
    fpv = oracle.encode_stream(nz, nz[0], 256, 256, 0, False)
    assert fpv.size == w["fpv_size"] and sha(fpv) == w["fpv_sha256"]


def test_plasma_1280x800_planes_and_stream(oracle, synth):
    w = MAN["workloads"]["plasma12_1280x800x20_seed7"]
    pl = synth.plasma_frames(20, 1280, 800, bits=12, seed=7)
    assert sha(pl) == w["input_sha256"]
    for k, want in enumerate(w["planes"]):
        h, l, pv, f = oracle.predict(pl[k], 1280, 800, 4, False, pl[0])
        assert (f, sha(h), sha(l), sha(pv)) == (want["flags"], want["high_sha256"],
                                                want["low_sha256"], want["preview_sha256"])
    fpv = oracle.encode_stream(pl, pl[0], 1280, 800, 4, False)
    assert fpv.size == w["fpv_size"] and sha(fpv) == w["fpv_sha256"]


def test_1920x1080_16bit_stream(oracle, synth):
    w = MAN["workloads"]["plasma16_1920x1080x4_seed9"]
    p = synth.plasma_frames(4, 1920, 1080, bits=16, seed=9)
    assert sha(p) == w["input_sha256"]
    fpv = oracle.encode_stream(p, p[0], 1920, 1080, 0, False)
    assert sha(fpv) == w["fpv_sha256"]
    assert np.array_equal(oracle.decode_stream(fpv), p)


def test_entropy_estimate_model(oracle):
    rng = np.random.default_rng(0)
    for _ in range(200):
        hist = rng.integers(0, 5000, 256).astype(np.uint64) * (rng.random(256) < rng.random())
        s = int(hist.sum())
        if s == 0:
            assert oracle.entropy(hist) == 0
            continue
        L = s.bit_length() - 1
        S = sum(int(v) * (L - (int(v).bit_length() - 1)) for v in hist if v)
        assert oracle.entropy(hist) == 1024 * S // s
    one = np.zeros(256, np.uint64)
    one[7] = 4370
    assert oracle.entropy(one) == 0                      # single bin: never "better than nothing"


def test_clamped_gradient_is_median(oracle):
    rng = np.random.default_rng(1)
    for n, w, nw in rng.integers(0, 256, (4000, 3)):
        n, w, nw = int(n), int(w), int(nw)
        assert oracle.cg(n, w, nw) == sorted((n, w, n + w - nw))[1]
    for n in (0, 1, 127, 128, 254, 255):
        for w in (0, 1, 127, 128, 254, 255):
            for nw in (0, 1, 127, 128, 254, 255):
                assert oracle.cg(n, w, nw) == sorted((n, w, n + w - nw))[1]


def test_flat_neighbourhood_wraps_rows(oracle):
    # column 0's west neighbour is the previous row's last pixel (flat 1-D indexing)
    xs, ys = 8, 4
    high = np.arange(xs * ys, dtype=np.uint16).reshape(ys, xs) * 7 % 251
    img = (high << 8).astype(np.uint16)
    h, _, _, f = oracle.predict(img, xs, ys, 0, False, None)
    flat = (img >> 8).astype(np.int64).reshape(-1)
    if f & 2:
        i = 2 * xs                                       # pixel (2, 0)
        pred = sorted((flat[i - xs], flat[i - 1], flat[i - xs] + flat[i - 1] - flat[i - xs - 1]))[1]
        assert h[i] == (flat[i] - pred) % 256
        assert np.array_equal(h[:xs + 1], flat[:xs + 1].astype(np.uint8))   # literals


def test_oracle_matches_reference_live(oracle):
    from oracle.oracle import Ref
    if not Ref.available():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    ref = Ref()
    rng = np.random.default_rng(5)
    for trial in range(12):
        xs, ys = int(rng.integers(1, 12)) * 4, int(rng.integers(1, 12)) * 4
        shift = int(rng.choice([0, 1, 4, 7, 8]))
        be = bool(rng.integers(0, 2))
        kind = trial % 3
        fr = rng.integers(0, 65536, (3, ys, xs)).astype(np.uint16)
        if kind == 1:
            fr = (fr >> 6) + 20000
        if kind == 2:
            fr = (np.cumsum(fr >> 12, axis=2) + 512).astype(np.uint16)
        for k in range(3):
            a = oracle.predict(fr[k], xs, ys, shift, be, fr[0])
            b = ref.predict(fr[k], xs, ys, shift, be, fr[0])
            assert a[3] == b[3] and np.array_equal(a[0], b[0]) and np.array_equal(a[2], b[2])
            if shift != 8:
                assert np.array_equal(a[1], b[1])
        assert np.array_equal(oracle.encode_stream(fr, fr[0], xs, ys, shift, be),
                              ref.encode_stream(fr, fr[0], xs, ys, shift, be, threads=2))
