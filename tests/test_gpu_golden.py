"""GPU path against the golden fixtures produced by the REAL reference
(tests/golden): predicted planes, flags, .fpv bytes, decoded frames, previews."""
import hashlib
import importlib
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = np.load(os.path.join(HERE, "golden", "edge_cases.npz"))
MAN = json.load(open(os.path.join(HERE, "golden", "manifest.json")))
NAMES = [str(n) for n in GOLD["names"]]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def codec(fpv):
    return importlib.import_module("fpv_amd.codec").Codec()


@pytest.mark.parametrize("name", NAMES)
def test_edge_case_planes_and_stream(fpv, codec, name):
    import torch
    frames, delta = GOLD[name + "/frames"], GOLD[name + "/delta"]
    xs, ys, shift, be = (int(v) for v in GOLD[name + "/meta"])
    n = frames.shape[0]
    p = fpv.Predictor(xs, ys, shift, bool(be), max_batch=n)
    p.set_delta_raw(delta)
    d = torch.from_numpy(np.ascontiguousarray(frames).reshape(n, -1).view(np.int16)).cuda().view(torch.uint16)
    high, low, pv, flags = p.predict(d)
    torch.cuda.synchronize()
    assert np.array_equal(flags.cpu().numpy(), GOLD[name + "/flags"])
    assert np.array_equal(high.cpu().numpy(), GOLD[name + "/high"])
    if shift != 8:
        assert np.array_equal(low.cpu().numpy(), GOLD[name + "/low"])
    assert np.array_equal(pv.cpu().numpy(), GOLD[name + "/preview"])
    # the delta frame itself: predicted without delta frame
    dd = torch.from_numpy(np.ascontiguousarray(delta).reshape(1, -1).view(np.int16)).cuda().view(torch.uint16)
    h0, _, _, f0 = p.predict(dd, with_delta=False)
    torch.cuda.synchronize()
    assert int(f0[0]) == int(GOLD[name + "/delta_flags"][0])
    assert np.array_equal(h0[0].cpu().numpy(), GOLD[name + "/delta_high"])
    p.close()
    fpv_bytes = codec.encode_stream(frames, delta, xs, ys, shift, bool(be), threads=2)
    assert np.array_equal(fpv_bytes, GOLD[name + "/fpv"])
    dec, dpv = codec.decode_stream(GOLD[name + "/fpv"], with_previews=True)
    assert np.array_equal(dec, GOLD[name + "/decoded"])
    assert np.array_equal(dpv, GOLD[name + "/decoded_preview"])
    assert np.array_equal(codec.stream_decode(GOLD[name + "/fpv"], xs, ys, n, block=97), GOLD[name + "/decoded"])


def test_config0_noise_stream_sha(codec, synth):
    w = MAN["workloads"]["noise_256x256x64_seed1"]
    nz = synth.noise_frames(64, 256, 256, seed=1)
    fpv_bytes = codec.encode_stream(nz, nz[0], 256, 256, 0, False, threads=4)
    assert fpv_bytes.size == w["fpv_size"] and sha(fpv_bytes) == w["fpv_sha256"]


def test_config1_plasma_stream_sha(codec, synth):
    w = MAN["workloads"]["plasma12_1280x800x20_seed7"]
    pl = synth.plasma_frames(20, 1280, 800, bits=12, seed=7)
    fpv_bytes = codec.encode_stream(pl, pl[0], 1280, 800, 4, False, threads=8)
    assert sha(fpv_bytes) == w["fpv_sha256"]
    assert np.array_equal(codec.decode_stream(fpv_bytes, batched=True, threads=4), pl << 4)


def test_config2_1080p_stream_sha(codec, synth):
    w = MAN["workloads"]["plasma16_1920x1080x4_seed9"]
    p16 = synth.plasma_frames(4, 1920, 1080, bits=16, seed=9)
    fpv_bytes = codec.encode_stream(p16, p16[0], 1920, 1080, 0, False, threads=4)
    assert sha(fpv_bytes) == w["fpv_sha256"]
    assert np.array_equal(codec.decode_stream(fpv_bytes), p16)


def test_config3_4k_stream_sha(codec, synth):
    w = MAN["workloads"]["plasma16_4096x4096x2_seed5"]
    p4k = synth.plasma_frames(2, 4096, 4096, bits=16, seed=5)
    fpv_bytes = codec.encode_stream(p4k, p4k[0], 4096, 4096, 0, False, threads=2)
    assert sha(fpv_bytes) == w["fpv_sha256"]
    assert np.array_equal(codec.decode_stream(fpv_bytes, batched=True, threads=2), p4k)


def test_full_size_roundtrip_property(fpv, synth):
    """BASELINE size, size-independent property: predict -> inverse == identity
    (on the shifted pixels) for a 256-frame resident batch, checked on the GPU."""
    import torch
    xs, ys, n = 1280, 800, 256
    base = synth.plasma_frames(16, xs, ys, bits=12, seed=3)
    p = fpv.Predictor(xs, ys, 4, False, max_batch=n)
    p.set_delta_raw(base[0])
    d = torch.from_numpy(base.reshape(16, -1).view(np.int16)).cuda()
    fr = d[torch.arange(n, device="cuda") % 16].contiguous().view(torch.uint16)
    high, low, pv, flags = p.predict(fr)
    from oracle.oracle import Oracle
    dh, dl, _ = Oracle().split(base[0], 4, False)
    p.set_delta_decoded((dh.astype(np.uint16) << 8) | dl)
    rec = p.inverse(high, low, flags)
    torch.cuda.synchronize()
    assert bool((rec.view(torch.int16).view(n, -1) == (fr.view(torch.int16).view(n, -1) << 4)).all())
    assert set(np.unique(flags.cpu().numpy())) <= {1, 3}
    p.close()
