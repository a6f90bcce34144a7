"""Host codec logic (container format, batching, worker pool, callback order,
decoder parsing) WITHOUT a GPU: the same host sources linked against the
oracle-backed stand-in for the C ABI (tests/shim).  Compared with golden
streams produced by the real reference."""
import hashlib
import importlib
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SHIM = os.environ.get("FPV_TEST_SHIM") or os.path.join(HERE, "shim", "libfpv_amd_cpushim.so")
GOLD = np.load(os.path.join(HERE, "golden", "edge_cases.npz"))
MAN = json.load(open(os.path.join(HERE, "golden", "manifest.json")))
NAMES = [str(n) for n in GOLD["names"]]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def codec(fpv):
    if not os.path.exists(SHIM):
        pytest.skip("tests/shim/libfpv_amd_cpushim.so not built (make shim)")
    return importlib.import_module("fpv_amd.codec").Codec(SHIM)


@pytest.mark.parametrize("name", NAMES)
@pytest.mark.parametrize("threads", [0, 3])
def test_edge_streams_match_reference(codec, name, threads):
    frames, delta = GOLD[name + "/frames"], GOLD[name + "/delta"]
    xs, ys, shift, be = (int(v) for v in GOLD[name + "/meta"])
    fpv = codec.encode_stream(frames, delta, xs, ys, shift, bool(be), threads=threads)
    assert np.array_equal(fpv, GOLD[name + "/fpv"])
    dec, pv = codec.decode_stream(fpv, with_previews=True)
    assert np.array_equal(dec, GOLD[name + "/decoded"])
    assert np.array_equal(pv, GOLD[name + "/decoded_preview"])
    assert np.array_equal(codec.decode_stream(fpv, batched=True, threads=2), GOLD[name + "/decoded"])
    for block in (1, 13, 4096, 1 << 20):
        assert np.array_equal(codec.stream_decode(fpv, xs, ys, frames.shape[0], block=block),
                              GOLD[name + "/decoded"])


@pytest.mark.parametrize("threads", [0, 1, 4, 16])
def test_noise_config0_stream(codec, synth, threads):
    w = MAN["workloads"]["noise_256x256x64_seed1"]
    nz = synth.noise_frames(64, 256, 256, seed=1)
    fpv = codec.encode_stream(nz, nz[0], 256, 256, 0, False, threads=threads)
    assert fpv.size == w["fpv_size"] and sha(fpv) == w["fpv_sha256"]


def test_plasma_stream_and_decoders(codec, synth):
    w = MAN["workloads"]["plasma12_1280x800x20_seed7"]
    pl = synth.plasma_frames(20, 1280, 800, bits=12, seed=7)
    fpv = codec.encode_stream(pl, pl[0], 1280, 800, 4, False, threads=8)
    assert sha(fpv) == w["fpv_sha256"]
    assert np.array_equal(codec.decode_stream(fpv, batched=True, threads=4), pl << 4)
    assert np.array_equal(codec.stream_decode(fpv, 1280, 800, 20, block=1 << 20), pl << 4)
    raw = codec.unextract(codec.decode_stream(fpv)[7], 1280, 800, 4, False)
    assert np.array_equal(raw.view(np.uint16), pl[7].reshape(-1))


def test_callbacks_in_order_with_payloads(codec, synth):
    fr = synth.noise_frames(50, 64, 48, seed=2)
    for threads in (0, 2, 9):
        mq, got = codec.encode_with_callbacks(fr, fr[0], 64, 48, threads=threads)
        assert mq == (1 if threads == 0 else mq) and mq >= max(1, threads)
        assert [g[0] for g in got] == [1000000] + list(range(1, 51)) + [2000000]   # Finish twice = once
        stream = np.frombuffer(b"".join(g[1] for g in got), np.uint8)
        assert np.array_equal(codec.decode_stream(stream), fr)
        # frame index offsets point at the chunks in emission order
        n = 50
        foot = stream[-(5 + 8 * n + 8):]
        offs = np.frombuffer(foot[5:5 + 8 * n].tobytes(), "<u8")
        pos = len(got[0][1])
        for k in range(n):
            assert offs[k] == pos
            pos += len(got[1 + k][1])


def test_zero_frames_stream(codec, synth):
    fr = synth.noise_frames(1, 64, 48, seed=4)
    fpv = codec.encode_stream(fr[:0], fr[0], 64, 48, threads=2)
    assert codec.stream_info(fpv) == (64, 48, 0)
    assert codec.decode_stream(fpv).shape[0] == 0


def test_corrupt_streams_fail_cleanly(codec, capfd):
    name = "plasma12_le_s4"
    good = GOLD[name + "/fpv"].copy()
    bad = good.copy()
    bad[12] = 0                                     # delta chunk flag must be 1
    with pytest.raises(RuntimeError):
        codec.decode_stream(bad)
    assert "failure at:" in capfd.readouterr().err  # same diagnostics style as the reference
    bad = good.copy()
    bad[-9] ^= 0xFF                                 # last frame offset in the index
    with pytest.raises(RuntimeError):
        codec.decode_stream(bad)
    bad = good.copy()
    bad[40:60] ^= 0x55                              # inside the delta frame's Brotli data
    with pytest.raises(RuntimeError):
        codec.decode_stream(bad)
    with pytest.raises(RuntimeError):
        codec.stream_decode(bad, 64, 48, 3, block=100)
    with pytest.raises(RuntimeError):
        codec.decode_stream(good[:-4])              # truncated footer
    # a truncated stream simply yields fewer frames from the streaming decoder
    assert codec.stream_decode(good[:good.size // 2], 64, 48, 3, block=777).shape[0] < 3


def test_streaming_decoder_whole_file_and_large_blocks(codec, synth):
    """Calls that carry several complete frames go through the batched path
    (threaded Brotli, GPU batches); results and order must equal the
    block-by-block feed, for every way of cutting the stream."""
    fr = synth.noise_frames(37, 64, 48, seed=11)
    fr[5:20] >>= 6                                   # a run of CG-friendly frames
    fpv = codec.encode_stream(fr, fr[0], 64, 48, threads=3)
    for block in (0, fpv.size // 2 + 1, fpv.size // 3, 20000, 6500, 64):
        assert np.array_equal(codec.stream_decode(fpv, 64, 48, 37, block=block), fr), block


def test_streaming_decoder_failure_order_in_batches(codec, synth, capfd):
    """A corrupt frame in the middle of a multi-frame call: the frames in front
    of it are delivered, in order, then ok=false -- as the reference's sequential
    loop does (cc:914-936)."""
    fr = synth.noise_frames(12, 64, 48, seed=12)
    fpv = codec.encode_stream(fr, fr[0], 64, 48, threads=2).copy()
    n = 12
    foot = fpv[-(5 + 8 * n + 8):]
    offs = np.frombuffer(foot[5:5 + 8 * n].tobytes(), "<u8").astype(np.int64)
    k = 7
    psz = int(np.frombuffer(fpv[offs[k] + 5:offs[k] + 9].tobytes(), "<u4")[0])
    main = int(offs[k]) + 9 + psz
    fpv[main + 1:main + 40] ^= 0xA5                  # inside frame 7's Brotli streams
    for block in (0, 30000):
        with pytest.raises(RuntimeError) as ei:
            codec.stream_decode(fpv, 64, 48, n, block=block)
        assert np.array_equal(ei.value.delivered, fr[:k]), block
    bad = codec.encode_stream(fr, fr[0], 64, 48, threads=2).copy()
    bad[offs[9] + 4] = 3                             # chunk type of frame 9: not a standard frame
    with pytest.raises(RuntimeError) as ei:
        codec.stream_decode(bad, 64, 48, n, block=0)
    assert np.array_equal(ei.value.delivered, fr[:9])
    capfd.readouterr()


def test_decode_frames_raw_equals_unextract(codec, oracle, synth):
    """DecodeFramesRaw (UnextractFrame fused behind the inverse) gives the bytes the
    host loop gives, for every shift / endianness the CLI accepts -- i.e. the file
    the frames came from."""
    for shift, be, bits in ((0, False, 16), (4, False, 12), (4, True, 12), (8, True, 8), (0, True, 16)):
        fr = synth.noise_frames(9, 64, 48, seed=30 + shift) >> (16 - bits)
        src = fr.byteswap() if be else fr
        fpv = codec.encode_stream(src, src[0], 64, 48, shift, be, threads=2)
        raw = codec.decode_stream_raw(fpv, shift, be, threads=3)
        assert np.array_equal(raw.reshape(-1), src.reshape(-1).view(np.uint8)), (shift, be)
        dec = codec.decode_stream(fpv)
        for k in (0, 8):
            assert np.array_equal(raw[k], oracle.unextract(dec[k], shift, be))


def test_unextract_matches_oracle(codec, oracle, synth):
    img = synth.noise_frames(1, 64, 48, seed=8)[0]
    for shift, be in ((0, False), (4, False), (4, True), (8, True)):
        assert np.array_equal(codec.unextract(img, 64, 48, shift, be), oracle.unextract(img, shift, be))


def test_in_process_multi_device_round_robin(tmp_path, synth):
    """Encoder shards batches over every visible device (batch b -> device b mod G)
    and RandomAccessDecoder::DecodeFrames shards its frame list the same way.
    The CPU shim can pose as several devices: stream and pixels must not change."""
    import subprocess
    import sys
    if not os.path.exists(SHIM):
        pytest.skip("tests/shim not built")
    code = r'''
import sys, os, importlib, hashlib
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
from conftest import load_package
load_package()
synth = importlib.import_module("fpv_amd.synth"); codec = importlib.import_module("fpv_amd.codec").Codec(sys.argv[2])
fr = synth.noise_frames(70, 64, 48, seed=9)
fpv = codec.encode_stream(fr, fr[0], 64, 48, threads=4)
print("SHA", hashlib.sha256(fpv.tobytes()).hexdigest())
# batched random-access decode: 70 frames, batches of 16 -> one shard per visible device
import numpy as np
dec = codec.decode_stream(fpv, batched=True, threads=4)
assert np.array_equal(dec, fr), "sharded DecodeFrames differs"
print("DECODE ok")
'''
    script = tmp_path / "multi.py"
    script.write_text(code)
    root = os.path.dirname(HERE)
    shas = []
    for ndev, devs in ((1, None), (3, None), (4, "1,3")):
        env = dict(os.environ, FPV_SHIM_DEVICES=str(ndev), FPV_DECODE_BATCH="16")
        if devs:
            env["FPV_DEVICES"] = devs
        out = subprocess.run([sys.executable, str(script), root, SHIM], env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stdout + out.stderr
        assert "DECODE ok" in out.stdout, out.stdout + out.stderr
        shas.append([l for l in out.stdout.splitlines() if l.startswith("SHA")][0])
    assert shas[0] == shas[1] == shas[2]


_PIPE_SCRIPT = r'''
import sys, os, importlib, hashlib, ctypes, time
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
from conftest import load_package
load_package()
synth = importlib.import_module("fpv_amd.synth"); codec = importlib.import_module("fpv_amd.codec").Codec(sys.argv[2])
fr = synth.noise_frames(int(sys.argv[3]), 64, 48, seed=9)
t0 = time.time()
mq, got = codec.encode_with_callbacks(fr, fr[0], 64, 48, threads=int(sys.argv[4]))
dt = time.time() - t0
ids = [int(p) for p, _ in got]
print("IDS", " ".join(map(str, ids)))
print("SHA", hashlib.sha256(b"".join(b for _, b in got)).hexdigest())
print("ERR", codec.last_error)
print("SECONDS %.3f" % dt)
lib = ctypes.CDLL(os.path.join(os.path.dirname(sys.argv[2]), "libfpv_amd_cpushim.so"))
if hasattr(lib, "fpvhip_shim_overlapped_waits"):
    print("OVERLAPPED", lib.fpvhip_shim_overlapped_waits())
'''


def _run_pipe(tmp_path, env, nframes=160, threads=4):
    import subprocess
    import sys
    script = tmp_path / "pipe.py"
    script.write_text(_PIPE_SCRIPT)
    out = subprocess.run([sys.executable, str(script), os.path.dirname(HERE), SHIM, str(nframes), str(threads)],
                         env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    return {l.split(" ", 1)[0]: l.split(" ", 1)[1] if " " in l else "" for l in out.stdout.splitlines() if l[:1].isupper()}, out.stderr


def test_slow_device_does_not_block_the_others(tmp_path):
    """Eight devices, one of them slow (cc:1217-1224 keeps the EMISSION in order; nothing says the
    hand-off of finished GPU batches has to be).  One dispatcher per device: while the slow
    device's fpvhip_predict_wait is still sleeping, the waits of the other seven return and their
    batches go to the Brotli workers.  A single FIFO dispatcher (round 2) would show zero such
    overlaps.  Emission order and bytes are those of one device."""
    if not os.path.exists(SHIM):
        pytest.skip("tests/shim not built")
    base, _ = _run_pipe(tmp_path, {"FPV_SHIM_DEVICES": "1", "FPV_BATCH": "4"})
    slow, _ = _run_pipe(tmp_path, {"FPV_SHIM_DEVICES": "8", "FPV_SHIM_DELAY": "3:40", "FPV_BATCH": "4"})
    assert slow["SHA"] == base["SHA"]
    ids = [int(v) for v in slow["IDS"].split()]
    assert ids == [1000000] + list(range(1, 161)) + [2000000]        # header, frames in order, index
    assert int(slow["OVERLAPPED"]) > 0
    assert slow["ERR"] == ""


def test_gpu_failure_is_latched_not_fatal(tmp_path):
    """A failing fpvhip_predict_wait in the middle of a session (a lost device): the process goes
    on, the frames of batches before the failure have been emitted in order, later ones are
    dropped, Finish emits NO index, Encoder::LastError tells why.  FPV_ABORT_ON_ERROR=1 keeps
    round 2's abort()."""
    import subprocess
    import sys
    if not os.path.exists(SHIM):
        pytest.skip("tests/shim not built")
    # (the 10th wait of the process: two warm-ups, the delta frame, then batches -- a device may run up to three batches ahead
    # of the other, so with a smaller number a loaded machine can make the failing wait batch 0's, an EMPTY prefix)
    res, err = _run_pipe(tmp_path, {"FPV_SHIM_DEVICES": "2", "FPV_SHIM_FAIL_WAIT": "10", "FPV_BATCH": "4"})
    ids = [int(v) for v in res["IDS"].split()]
    assert ids[0] == 1000000 and 2000000 not in ids and 3000000 not in ids   # header, but no index
    frames = ids[1:]
    assert frames == list(range(1, len(frames) + 1)) and 0 < len(frames) < 160   # a clean prefix, in order
    assert "injected failure" in res["ERR"] and "encoder session failed" in err
    # the old behaviour on request
    script = tmp_path / "pipe.py"
    out = subprocess.run([sys.executable, str(script), os.path.dirname(HERE), SHIM, "40", "4"],
                         env=dict(os.environ, FPV_SHIM_FAIL_WAIT="3", FPV_BATCH="4", FPV_ABORT_ON_ERROR="1"),
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "fatal" in out.stderr


def test_gpu_failure_at_random_points_leaves_a_clean_prefix(tmp_path):
    """The same failure injected at a random wait of sessions of random shape (devices, batch size, worker threads, frames): whatever
    the point -- a warm-up, the delta frame in Init (then not even the header comes out), any batch of any device -- the callbacks
    have seen the header and frames 1 .. k in order and nothing else, k < n and no index when the failure hit, everything and the
    index when the session ended before the injected wait was reached."""
    import random
    if not os.path.exists(SHIM):
        pytest.skip("tests/shim not built")
    rnd = random.Random(5)
    hit = clean = 0
    for _ in range(10):
        devs, fw = rnd.choice([1, 2, 3, 8]), rnd.randint(4, 40)
        batch, th, n = rnd.choice([1, 3, 4, 8]), rnd.choice([1, 2, 4, 8]), rnd.choice([40, 97])
        res, err = _run_pipe(tmp_path, {"FPV_SHIM_DEVICES": str(devs), "FPV_SHIM_FAIL_WAIT": str(fw), "FPV_BATCH": str(batch)}, nframes=n, threads=th)
        ids = [int(v) for v in res["IDS"].split()]
        frames = [i for i in ids if i < 1000000]
        what = dict(devs=devs, fail_wait=fw, batch=batch, threads=th, frames=n)
        assert frames == list(range(1, len(frames) + 1)), what
        if "injected failure" in res["ERR"]:
            hit += 1
            assert 2000000 not in ids and len(frames) < n and (not ids or ids[0] == 1000000), what
        else:
            clean += 1
            assert ids[0] == 1000000 and ids[-1] == 2000000 and len(frames) == n, what
    assert hit >= 3, (hit, clean)


_WARM_SCRIPT = r'''
import sys, os, importlib, hashlib, ctypes
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
from conftest import load_package
load_package()
synth = importlib.import_module("fpv_amd.synth"); codec = importlib.import_module("fpv_amd.codec").Codec(sys.argv[2])
lib = ctypes.CDLL(sys.argv[2])
ndev = int(os.environ["FPV_SHIM_DEVICES"])
print("LOADED", " ".join(str(lib.fpvhip_shim_warmups(d)) for d in range(ndev)))   # nothing at load
fr = synth.noise_frames(64, 64, 48, seed=9)
fpv = codec.encode_stream(fr, fr[0], 64, 48, threads=4)
print("SHA", hashlib.sha256(fpv.tobytes()).hexdigest())
print("AFTER_ENCODE", " ".join(str(lib.fpvhip_shim_warmups(d)) for d in range(ndev)))
print("COLD", lib.fpvhip_shim_cold_creates())
import numpy as np
dec = codec.decode_stream(fpv, batched=True, threads=4)
assert np.array_equal(dec, fr)
fpv2 = codec.encode_stream(fr, fr[0], 64, 48, threads=2)                            # a second session: nothing is warmed twice
assert np.array_equal(fpv2, fpv)
print("AFTER_ALL", " ".join(str(lib.fpvhip_shim_warmups(d)) for d in range(ndev)))
print("COLD_END", lib.fpvhip_shim_cold_creates())
'''


def test_every_device_is_warmed_once_before_its_first_context(tmp_path):
    """Eight devices (FPV_DEVICES unset = all): loading the library warms nothing; the first
    Encoder warms EVERY device for the forward direction (one fpvhip_warmup each) before any
    context is created on it; the first decoder adds one more bring-up per device for the inverse
    direction, and later sessions warm nothing again.  Reference analogue: the thread spawn in the
    Encoder constructor, cc:1076-1084."""
    import subprocess
    import sys
    if not os.path.exists(SHIM):
        pytest.skip("tests/shim not built")
    script = tmp_path / "warm.py"
    script.write_text(_WARM_SCRIPT)
    for devs, expect in ((None, range(8)), ("5,2,5", (2, 5))):
        env = dict(os.environ, FPV_SHIM_DEVICES="8", FPV_BATCH="4", FPV_DECODE_BATCH="16")
        env.pop("FPV_WARMUP", None)
        if devs:
            env["FPV_DEVICES"] = devs
        out = subprocess.run([sys.executable, str(script), os.path.dirname(HERE), SHIM], env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stdout + out.stderr
        res = {l.split(" ", 1)[0]: l.split(" ", 1)[1] for l in out.stdout.splitlines() if l[:1].isupper() and " " in l}
        assert res["LOADED"].split() == ["0"] * 8
        want = ["1" if d in expect else "0" for d in range(8)]
        assert res["AFTER_ENCODE"].split() == want, res
        assert res["COLD"] == "0" and res["COLD_END"] == "0", res
        assert res["AFTER_ALL"].split() == ["2" if d in expect else "0" for d in range(8)], res


_FAIL_THEN_OK_SCRIPT = r'''
import sys, os, importlib, hashlib
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
from conftest import load_package
load_package()
synth = importlib.import_module("fpv_amd.synth"); codec = importlib.import_module("fpv_amd.codec").Codec(sys.argv[2])
fr = synth.noise_frames(96, 64, 48, seed=9)
good = codec.encode_stream(fr, fr[0], 64, 48, threads=4)          # fills the context cache
os.environ["FPV_SHIM_FAIL_WAIT"] = "4"
mq, got = codec.encode_with_callbacks(fr, fr[0], 64, 48, threads=4)
print("ERR1", codec.last_error)
os.environ.pop("FPV_SHIM_FAIL_WAIT")
for k in range(3):                                                 # the failed session's contexts are gone, not cached
    again = codec.encode_stream(fr, fr[0], 64, 48, threads=4)
    assert hashlib.sha256(again.tobytes()).digest() == hashlib.sha256(good.tobytes()).digest(), k
other = codec.encode_stream(fr[:, :32, :], fr[0, :32, :].copy(), 64, 32, threads=4)   # another geometry in between
again = codec.encode_stream(fr, fr[0], 64, 48, threads=4)
assert hashlib.sha256(again.tobytes()).digest() == hashlib.sha256(good.tobytes()).digest()
print("ERR2", codec.last_error)
print("OK")
'''


def test_failed_session_then_more_sessions_in_one_process(tmp_path):
    """A session that latched a GPU failure destroys its contexts; they must also leave the
    process-wide cache (a dangling entry marked busy would otherwise stay there, and a new context
    at the same address would flip it to idle: two sessions on one context).  Sessions that
    follow in the same process, same and other geometry, two devices, produce the reference
    bytes."""
    import subprocess
    import sys
    if not os.path.exists(SHIM):
        pytest.skip("tests/shim not built")
    script = tmp_path / "fail_then_ok.py"
    script.write_text(_FAIL_THEN_OK_SCRIPT)
    out = subprocess.run([sys.executable, str(script), os.path.dirname(HERE), SHIM],
                         env=dict(os.environ, FPV_SHIM_DEVICES="2", FPV_BATCH="4"), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout + out.stderr
    assert "injected failure" in out.stdout.split("ERR1", 1)[1].splitlines()[0]


_STATS_SCRIPT = r'''
import sys, os, ctypes, importlib, numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
from conftest import load_package
load_package()
synth = importlib.import_module("fpv_amd.synth")
lib = ctypes.CDLL(sys.argv[2])
lib.fpvc_bench_encode.restype = ctypes.c_double
lib.fpvc_bench_encode.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                                  ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_double), ctypes.c_size_t]
fr = np.ascontiguousarray(synth.noise_frames(8, 64, 48, seed=3)).reshape(-1)
st, nb = (ctypes.c_double * 43)(), ctypes.c_size_t()
dt = lib.fpvc_bench_encode(fr.ctypes.data, 8, 120, 64, 48, 0, 0, 4, ctypes.byref(nb), st, 43)
print("STATS", " ".join("%g" % v for v in st))
'''


def test_encoder_stats_per_device(tmp_path):
    """Encoder::LastStats (extension): frames, batches and, per device of the session, the batches
    it took and the time its dispatcher waited for it -- what the multi-GPU bench line's e2e_multi
    reports.  Three devices (the shim), 120 frames in batches of up to 4: the batches are dealt round robin
    (batch b goes to device b mod 3, cc:1199-1230 is the reference's own distribution axis)."""
    import subprocess
    import sys
    if not os.path.exists(SHIM):
        pytest.skip("tests/shim not built")
    script = tmp_path / "stats.py"
    script.write_text(_STATS_SCRIPT)
    out = subprocess.run([sys.executable, str(script), os.path.dirname(HERE), SHIM],
                         env=dict(os.environ, FPV_SHIM_DEVICES="3", FPV_BATCH="4"), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    st = [float(v) for v in [l for l in out.stdout.splitlines() if l.startswith("STATS")][0].split()[1:]]
    assert st[0] == 120 and st[10] == 3 and 30 <= st[5] <= 120     # (partial batches are flushed when the producer blocks or idles)
    per_dev = [st[11], st[13], st[15]]
    assert sum(per_dev) == st[5] and max(per_dev) - min(per_dev) <= 1, per_dev      # round robin
    assert all(st[12 + 2 * d] >= 0 for d in range(3)) and st[17] == 0               # nothing beyond the session's devices


def test_brotli_arena_bytes_are_the_librarys():
    """The host compresses every plane with libbrotli's state in a per-thread arena (round 4: the
    one-shot entry point's malloc / mmap / munmap per plane made sixteen worker threads queue on
    the address-space lock).  Same library, same parameters (cc:653: quality 1, default window,
    default mode), so the bytes must be those of BrotliEncoderCompress itself -- checked here call
    by call on empty, tiny, constant, random, preview-sized, plane-sized and 16 MB inputs, from
    several threads at once (each with its own arena)."""
    import ctypes
    import threading
    if not os.path.exists(SHIM):
        pytest.skip("tests/shim not built")
    lib = ctypes.CDLL(SHIM)
    lib.fpvc_test_brotli_equivalence.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    rng = np.random.default_rng(5)
    smooth = (np.add.outer(np.arange(800) // 3, np.arange(1280) // 5) + rng.integers(0, 3, (800, 1280))).astype(np.uint8).reshape(-1)
    cases = [np.zeros(0, np.uint8), np.zeros(1, np.uint8), rng.integers(0, 256, 100, dtype=np.uint8), np.zeros(64000, np.uint8),
             rng.integers(0, 256, 64000, dtype=np.uint8), smooth, rng.integers(0, 256, 1024000, dtype=np.uint8),
             (rng.integers(0, 4, 1 << 20) * 7).astype(np.uint8), smooth[: 4096 * 100], np.tile(smooth, 16),
             GOLD["plasma12_le_s4/fpv"].copy()]
    bad = []

    def run(offset):
        for k in range(len(cases)):
            c = np.ascontiguousarray(cases[(k + offset) % len(cases)])
            if lib.fpvc_test_brotli_equivalence(c.ctypes.data if c.size else None, c.size) != 1:
                bad.append((offset, c.size))
    th = [threading.Thread(target=run, args=(i,)) for i in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not bad, bad


def test_fuzzed_streams_never_crash(codec):
    """Random byte flips, truncations and splices: every decoder entry either
    succeeds or reports failure; none may crash or read out of bounds."""
    good = GOLD["plasma12_le_s4/fpv"].copy()
    rng = np.random.default_rng(1234)
    outcomes = [0, 0]
    for it in range(300):
        bad = good.copy()
        kind = it % 3
        if kind == 0:
            for _ in range(int(rng.integers(1, 4))):
                bad[int(rng.integers(0, bad.size))] ^= np.uint8(1 << int(rng.integers(0, 8)))
        elif kind == 1:
            bad = bad[:int(rng.integers(0, bad.size))]
        else:
            a, b = sorted(int(v) for v in rng.integers(0, bad.size, 2))
            bad = np.concatenate([bad[:a], bad[b:]])
        for fn in (lambda: codec.decode_stream(bad), lambda: codec.stream_decode(bad, 64, 48, 3, block=97),
                   lambda: codec.stream_info(bad)):
            try:
                fn()
                outcomes[0] += 1
            except RuntimeError:
                outcomes[1] += 1
    assert outcomes[0] > 0 and outcomes[1] > 0
