"""GPU parity of the whole host codec (Encoder / decoders over the HIP path)
against the CPU oracle: compressed bytes and decoded pixels must be identical."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def codec(fpv):
    import importlib
    return importlib.import_module("fpv_amd.codec").Codec()


@pytest.mark.parametrize("threads", [0, 1, 3, 8])
def test_noise_stream_bytes(codec, synth, oracle, threads):
    fr = synth.noise_frames(24, 256, 256, seed=1)
    want = oracle.encode_stream(fr, fr[0], 256, 256, 0, False)
    got = codec.encode_stream(fr, fr[0], 256, 256, 0, False, threads=threads)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("shift,be", [(4, False), (0, False), (4, True), (8, False)])
def test_plasma_stream_bytes_and_roundtrip(codec, synth, oracle, shift, be):
    xs, ys = 320, 200
    fr = synth.plasma_frames(12, xs, ys, bits=12 if shift else 16)
    if shift == 8:
        fr = fr >> 4                      # 8-bit data
    if be:
        fr = fr.byteswap()
    want = oracle.encode_stream(fr, fr[0], xs, ys, shift, be)
    got = codec.encode_stream(fr, fr[0], xs, ys, shift, be, threads=4)
    assert np.array_equal(got, want)
    dec = codec.decode_stream(got)
    assert np.array_equal(dec, oracle.decode_stream(want))
    raw = codec.unextract(dec[3], xs, ys, shift, be)
    assert np.array_equal(raw, fr[3].reshape(-1).view(np.uint8))
    assert np.array_equal(codec.decode_stream(got, batched=True, threads=3), dec)
    assert np.array_equal(codec.stream_decode(got, xs, ys, 12, block=4093), dec)


def test_full_size_frames(codec, synth, oracle):
    fr = synth.plasma_frames(6, 1280, 800, bits=12)
    want = oracle.encode_stream(fr, fr[0], 1280, 800, 4, False)
    got = codec.encode_stream(fr, fr[0], 1280, 800, 4, False, threads=6)
    assert np.array_equal(got, want)
    dec, pv = codec.decode_stream(got, with_previews=True)
    assert np.array_equal(dec, fr << 4)
    for k in range(6):
        h, _, _ = oracle.split(fr[k], 4, False)
        box = h.reshape(200, 4, 320, 4).astype(np.uint32).sum(axis=(1, 3))
        assert np.array_equal(pv[k], ((box // 16) & 0xfe).astype(np.uint8))


def test_callback_order_and_payloads(codec, synth):
    fr = synth.noise_frames(40, 64, 48, seed=5)
    mq, got = codec.encode_with_callbacks(fr, fr[0], 64, 48, threads=7)
    assert mq >= 7
    ids = [g[0] for g in got]
    assert ids == [1000000] + list(range(1, 41)) + [2000000]


@pytest.mark.parametrize("xs,ys", [(30, 22), (5, 3), (1, 7), (257, 9), (1284, 6), (1, 1), (3, 1), (1, 2)])
def test_decoders_accept_odd_sizes(codec, oracle, xs, ys):
    """Files written by the CPU reference may have any size (only ENCODING needs
    multiples of 4): the byte-granular inverse kernel decodes them."""
    rng = np.random.default_rng(xs * 1000 + ys)
    fr = (np.cumsum(rng.integers(0, 40, (4, ys, xs)), axis=2) * 37 + rng.integers(0, 300, (4, ys, xs))).astype(np.uint16)
    stream = oracle.encode_stream(fr, fr[0], xs, ys, 0, False)
    want = oracle.decode_stream(stream)
    assert np.array_equal(want, fr)
    assert np.array_equal(codec.decode_stream(stream), want)
    assert np.array_equal(codec.decode_stream(stream, batched=True, threads=2), want)
    assert np.array_equal(codec.stream_decode(stream, xs, ys, 4, block=50), want)
    if xs >= 4 and ys >= 4:
        _, pv = codec.decode_stream(stream, with_previews=True)
        assert pv.shape == (4, ys // 4, xs // 4)


def test_streaming_decoder_batched_calls(codec, synth):
    """StreamingDecoder fed whole files or large blocks takes the GPU-batch path
    (several frames per inverse launch, threaded Brotli): same pixels and order
    as the block-wise feed, at the BASELINE frame size."""
    fr = synth.plasma_frames(20, 1280, 800, bits=12, seed=7)
    fpv = codec.encode_stream(fr, fr[0], 1280, 800, 4, False, threads=8)
    want = fr << 4
    for block in (0, fpv.size // 3 + 7, 1 << 20):
        assert np.array_equal(codec.stream_decode(fpv, 1280, 800, 20, block=block), want), block
    # a corrupt frame inside a batch: everything in front of it is delivered first
    bad = fpv.copy()
    foot = bad[-(5 + 8 * 20 + 8):]
    offs = np.frombuffer(foot[5:5 + 8 * 20].tobytes(), "<u8").astype(np.int64)
    psz = int(np.frombuffer(bad[offs[11] + 5:offs[11] + 9].tobytes(), "<u4")[0])
    main = int(offs[11]) + 9 + psz
    bad[main + 1:main + 64] ^= 0x5A
    with pytest.raises(RuntimeError) as ei:
        codec.stream_decode(bad, 1280, 800, 20, block=0)
    assert np.array_equal(ei.value.delivered, want[:11])


@pytest.mark.parametrize("shift,be", [(4, False), (4, True), (0, True), (8, False), (0, False)])
def test_decode_frames_raw_fused_unextract(codec, synth, shift, be):
    """K5: UnextractFrame (cc:850-862) fused on the GPU behind the inverse --
    DecodeFramesRaw returns the bytes of the raw file the stream was made from."""
    xs, ys = 320, 200
    fr = synth.plasma_frames(10, xs, ys, bits=12 if shift == 4 else 16, seed=21)
    if shift == 8:
        fr = fr >> 8
    src = fr.byteswap() if be else fr
    fpv = codec.encode_stream(src, src[0], xs, ys, shift, be, threads=4)
    raw = codec.decode_stream_raw(fpv, shift, be, threads=3)
    assert np.array_equal(raw.reshape(-1), src.reshape(-1).view(np.uint8))
    dec = codec.decode_stream(fpv)
    assert np.array_equal(raw[3], codec.unextract(dec[3], xs, ys, shift, be))


def test_decoder_accepts_very_wide_frames(codec, oracle):
    """Decoders take any geometry the container can hold (up to 65536 wide): beyond the
    LDS budget of the speculative inverse kernels the serial-chain kernels take over."""
    for xs, ys in ((50000, 2), (65536, 1)):
        rng = np.random.default_rng(xs)
        fr = (np.cumsum(rng.integers(0, 9, (2, ys, xs)), axis=2) * 5 + rng.integers(0, 50, (2, ys, xs))).astype(np.uint16)
        stream = oracle.encode_stream(fr, fr[0], xs, ys, 0, False)
        assert np.array_equal(codec.decode_stream(stream), fr)
        assert np.array_equal(codec.decode_stream(stream, batched=True, threads=2), fr)


@pytest.mark.parametrize("xs,ys", [(8, 4096), (4, 16384), (8192, 64), (4096, 260), (20, 1000)])
def test_extreme_aspect_ratios(codec, oracle, xs, ys):
    """Tall-and-narrow and wide-and-short frames through the whole codec: stream bytes
    identical to the oracle's, both decoders return the pixels."""
    rng = np.random.default_rng(xs * 31 + ys)
    fr = (np.cumsum(rng.integers(0, 12, (3, ys, xs)), axis=1) * 3 + rng.integers(0, 90, (3, ys, xs)) + 500).astype(np.uint16)
    want = oracle.encode_stream(fr, fr[0], xs, ys, 0, False)
    got = codec.encode_stream(fr, fr[0], xs, ys, 0, False, threads=3)
    assert np.array_equal(got, want)
    assert np.array_equal(codec.decode_stream(got), fr)
    assert np.array_equal(codec.stream_decode(got, xs, ys, 3, block=0), fr)


def test_concurrent_sessions_in_one_process(codec, synth):
    """Several encoder / decoder sessions at once from different threads (ctypes drops
    the GIL during the calls): contexts come from the process-wide pool, kernels run on
    independent streams; every session must produce what it produces alone."""
    import threading
    jobs = []
    for seed, (xs, ys, shift) in enumerate([(320, 200, 4), (256, 256, 0), (320, 200, 4), (1280, 800, 4)]):
        fr = synth.plasma_frames(12, xs, ys, bits=12 if shift else 16, seed=40 + seed)
        jobs.append((fr, xs, ys, shift))
    alone = [codec.encode_stream(fr, fr[0], xs, ys, shift, False, threads=3) for fr, xs, ys, shift in jobs]
    results = [None] * len(jobs)
    decoded = [None] * len(jobs)

    def run(i):
        fr, xs, ys, shift = jobs[i]
        results[i] = codec.encode_stream(fr, fr[0], xs, ys, shift, False, threads=3)
        decoded[i] = codec.decode_stream(results[i], batched=True, threads=2)

    for _ in range(2):
        ts = [threading.Thread(target=run, args=(i,)) for i in range(len(jobs))]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        for i, (fr, xs, ys, shift) in enumerate(jobs):
            assert np.array_equal(results[i], alone[i]), f"session {i} differs when run concurrently"
            assert np.array_equal(decoded[i], fr << shift)


def test_two_contexts_on_one_device(codec, oracle, synth, monkeypatch):
    """FPV_DEVICES=0,0: the in-process multi-GPU paths on real HIP -- two forward contexts (own
    slots, own stream sets) feeding one Encoder round-robin, and DecodeFrames sharded over two
    inverse engines.  Bytes and pixels identical to the single-context run (cc:1179-1183: the
    only serial point is the in-order callback + index)."""
    fr = synth.plasma_frames(70, 320, 200, bits=12, seed=11)
    one = codec.encode_stream(fr, fr[0], 320, 200, 4, False, threads=6)
    assert np.array_equal(one, oracle.encode_stream(fr, fr[0], 320, 200, 4, False))
    monkeypatch.setenv("FPV_DEVICES", "0,0")
    monkeypatch.setenv("FPV_BATCH", "4")
    monkeypatch.setenv("FPV_DECODE_BATCH", "8")
    two = codec.encode_stream(fr, fr[0], 320, 200, 4, False, threads=6)
    assert np.array_equal(one, two)
    dec = codec.decode_stream(two, batched=True, threads=6)      # 70 frames / batch 8: two shards
    assert np.array_equal(dec, fr << 4)
    assert np.array_equal(codec.decode_stream_raw(two, shift=4, threads=6).view(np.uint16).reshape(70, 200, 320), fr)
