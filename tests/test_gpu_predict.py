"""GPU parity of the forward / inverse predictor kernels (through the C ABI,
device-resident entry points) against the CPU oracle on the same inputs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run_case(fpv, oracle, frames, xs, ys, shift, be, delta_raw):
    import torch
    n = frames.shape[0]
    p = fpv.Predictor(xs, ys, shift, be, max_batch=max(n, 1))
    if delta_raw is not None:
        p.set_delta_raw(delta_raw)
    d = torch.from_numpy(frames.reshape(n, -1).view(np.int16)).cuda().view(torch.uint16)
    high, low, pv, flags = p.predict(d, with_delta=delta_raw is not None)
    torch.cuda.synchronize()
    high, low, pv, flags = (t.cpu().numpy() for t in (high, low, pv, flags))
    for k in range(n):
        eh, el, ep, ef = oracle.predict(frames[k], xs, ys, shift, be, delta_raw)
        assert flags[k] == ef, f"frame {k}: flags {flags[k]} != {ef}"
        assert np.array_equal(high[k], eh), f"frame {k}: high plane differs"
        if shift != 8:
            assert np.array_equal(low[k], el), f"frame {k}: low plane differs"
        assert np.array_equal(pv[k], ep), f"frame {k}: preview differs"
    # inverse: back to the (shifted) pixels
    if delta_raw is not None:
        dh, dl, _ = oracle.split(delta_raw, shift, be)
        p.set_delta_decoded((dh.astype(np.uint16) << 8) | dl)
    th, tl, tf = (torch.from_numpy(a).cuda() for a in (high, low, flags))
    rec = p.inverse(th, tl, tf)
    torch.cuda.synchronize()
    rec = rec.cpu().view(torch.int16).numpy().view(np.uint16)
    for k in range(n):
        delta16 = None if delta_raw is None else ((dh.astype(np.uint16) << 8) | dl)
        exp = oracle.inverse(high[k], None if flags[k] & 4 else low[k], delta16, xs, ys, int(flags[k]))
        assert np.array_equal(rec[k], exp), f"frame {k}: inverse differs"
    p.close()
    return flags


@pytest.mark.parametrize("shift,be", [(0, False), (4, False), (0, True), (4, True), (8, False), (8, True)])
def test_noise_256(fpv, synth, oracle, shift, be):
    fr = synth.noise_frames(8, 256, 256, seed=1)
    _run_case(fpv, oracle, fr, 256, 256, shift, be, fr[0])


def test_plasma_1280x800(fpv, synth, oracle):
    fr = synth.plasma_frames(4, 1280, 800, bits=12)
    flags = _run_case(fpv, oracle, fr, 1280, 800, 4, False, fr[0])
    assert list(flags[1:]) == [3, 3, 3]


def test_no_delta_frame(fpv, synth, oracle):
    fr = synth.plasma_frames(2, 1280, 800, bits=12)
    _run_case(fpv, oracle, fr, 1280, 800, 4, False, None)


@pytest.mark.parametrize("xs,ys", [(64, 48), (4, 4), (8, 4), (4096, 8), (1920, 1080), (36, 20), (6144, 8), (8192, 4), (8196, 4), (16384, 8), (32768, 4)])
def test_shapes(fpv, synth, oracle, xs, ys):
    fr = synth.noise_frames(3, xs, ys, seed=3)
    fr[1] = (fr[1] >> 9) + 3000          # smooth-ish: exercises the CG-on branch
    _run_case(fpv, oracle, fr, xs, ys, 0, False, fr[0])


def test_partial_batches_reuse_workspace(fpv, synth, oracle):
    """A context sized for 8 frames is reused for smaller batches whose flags
    differ (zero low plane after non-zero, CG on after CG off): per-batch state
    (NO_LOW accumulators, redo lists, speculation) must not leak."""
    import torch
    xs, ys = 256, 64
    noise = synth.noise_frames(8, xs, ys, seed=21)
    smooth = ((synth.plasma_frames(3, xs, ys, bits=12, seed=2) << 4) & 0xFF00).astype(np.uint16)
    p = fpv.Predictor(xs, ys, 0, False, max_batch=8)
    p.set_delta_raw(noise[0])
    for frames in (noise, smooth, noise[:5], smooth[:1], noise[2:3]):
        n = frames.shape[0]
        d = torch.from_numpy(np.ascontiguousarray(frames).reshape(n, -1).view(np.int16)).cuda().view(torch.uint16)
        high, low, pv, flags = p.predict(d)
        torch.cuda.synchronize()
        for k in range(n):
            eh, el, ep, ef = oracle.predict(frames[k], xs, ys, 0, False, noise[0])
            assert int(flags[k]) == ef
            assert np.array_equal(high[k].cpu().numpy(), eh)
            assert np.array_equal(low[k].cpu().numpy(), el)
            assert np.array_equal(pv[k].cpu().numpy(), ep)
    p.close()


@pytest.mark.parametrize("env", [{"FPV_FORWARD": "stream", "FPV_STRIP_ROWS": "4"}, {"FPV_FORWARD": "stream", "FPV_STRIP_ROWS": "80", "FPV_STRIP_MAJOR": "0"},
                                 {"FPV_FORWARD": "stream", "FPV_STRIP_ROWS": "12", "FPV_WAVES_PER_CU": "2"}, {"FPV_FORWARD": "stream"},
                                 {"FPV_TILE_GROUPS": "1"}, {"FPV_TILE_GROUPS": "7"}, {"FPV_TILE_GROUPS": "50"}, {"FPV_FORWARD": "tile16"},
                                 {"FPV_TILE_FGROUP": "2"}, {"FPV_TILE_FGROUP": "4", "FPV_TILE_GROUPS": "7"}, {"FPV_TILE_FGROUP": "1", "FPV_TILE_GROUPS": "1"},
                                 {"FPV_FORWARD": "tile8"}, {"FPV_FORWARD": "tile8", "FPV_FRAME8_BIG": "1"},
                                 {"FPV_FORWARD": "tile8", "FPV_TILE8_GROUPS": "1"}, {"FPV_FORWARD": "tile8", "FPV_TILE8_GROUPS": "2"},
                                 {"FPV_FORWARD": "tile8", "FPV_TILE8_GROUPS": "5"}, {"FPV_FORWARD": "tile8", "FPV_TILE8_GROUPS": "50"},
                                 {"FPV_FORWARD": "tile8", "FPV_TILE8_ROWS": "8"}, {"FPV_FORWARD": "tile8", "FPV_TILE8_ROWS": "8", "FPV_TILE8_GROUPS": "1"},
                                 {"FPV_FORWARD": "tile8", "FPV_TILE8_ROWS": "8", "FPV_TILE8_GROUPS": "7"},
                                 {"FPV_FORWARD": "tile8", "FPV_TILE8_ROWS": "8", "FPV_TILE8_GROUPS": "50"},
                                 {"FPV_INVERSE_SLOW": "1"}, {"FPV_INVERSE": "fast"}, {"FPV_INVERSE": "spec"},
                                 {"FPV_INVERSE": "trio"}])
def test_alternate_kernel_paths(fpv, synth, oracle, env):
    """The streaming forward kernels (round 2; still the path of widths that are not multiples
    of 64) with odd strip sizes and the frame-major task order, the tile pass (16 pixels per lane)
    and round 4's variant of it (FPV_FORWARD=tile8: 8 pixels per lane, 4 or 8 rows per thread) with
    other tile heights (one group per tile, a last tile that is partly outside the frame or ends on
    half a group, the whole frame as one tile) and every inverse kernel (byte-granular, serial chain, one-wave and three-wave
    speculative) must give the same bits as the defaults."""
    import os
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        fr = np.concatenate([synth.plasma_frames(3, 320, 200, bits=12, seed=5),
                             (synth.noise_frames(3, 320, 200, seed=6) >> 4)])
        _run_case(fpv, oracle, fr, 320, 200, 4, False, fr[0])
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("xs,ys,n,env,want", [
    (512, 40, 6, {"FPV_TILE_FGROUP": "2"}, 2), (1280, 48, 5, {"FPV_TILE_FGROUP": "3", "FPV_TILE_GROUPS": "5"}, 3), (4096, 24, 6, {"FPV_TILE_FGROUP": "1"}, 1),
    (4096, 1032, 3, {}, 6), (2048, 1536, 2, {}, 0),
    (512, 40, 6, {"FPV_TILE_FGROUP": "2", "FPV_TILE_XBAND": "1"}, 2), (1280, 48, 5, {"FPV_TILE_FGROUP": "3", "FPV_TILE_GROUPS": "1", "FPV_TILE_XBAND": "1"}, 3),
    (4096, 24, 6, {"FPV_TILE_FGROUP": "1", "FPV_TILE_XBAND": "1"}, 1), (320, 200, 7, {"FPV_TILE_FGROUP": "2", "FPV_TILE_GROUPS": "1", "FPV_TILE_XBAND": "1"}, 2),
    (320, 68, 3, {"FPV_TILE_FGROUP": "1", "FPV_TILE_GROUPS": "1", "FPV_TILE_XBAND": "1"}, 1), (4096, 1032, 3, {"FPV_TILE_XBAND": "0"}, 6)])
def test_large_frame_order(fpv, synth, oracle, xs, ys, n, env, want):
    """Frames whose delta planes pass 8 MB are walked in groups of frames TILE BY TILE (the frame of the
    group fastest; grid = (tiles << k, groups)) instead of frame by frame; forced here on small frames
    too (FPV_TILE_FGROUP=k) with batches that leave the last group short -- its empty slots are
    workgroups that leave at once -- and with a partial last tile.  Small batches of such frames go band-wise
    (FPV_TILE_XBAND: XCD c walks its own eighth of the tiles; by default up to 16 frames, so the 4096-wide cases here
    take it): forced on and off, with tile counts below 8, not a multiple of 8 (3, 12, 50, 17 tiles) and 129.
    Noise (the CG decision flips, frames are redone) and smooth frames, with and without a delta frame."""
    import os
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        p = fpv.Predictor(xs, ys, 4, False, max_batch=1)
        groups, threads, tiles, lfgroup = p.forward_tile_shape()
        p.close()
        assert lfgroup == want and tiles == -(-ys // (4 * groups)) and groups * (xs // 16) <= threads, (groups, threads, tiles, lfgroup)
        fr = np.concatenate([synth.plasma_frames(n - n // 2, xs, ys, bits=12, seed=5), (synth.noise_frames(n // 2, xs, ys, seed=6) >> 4)])
        _run_case(fpv, oracle, fr, xs, ys, 4, False, fr[0])
        _run_case(fpv, oracle, fr[1:], xs, ys, 4, False, None)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("xs,ys,n,env", [
    (320, 200, 6, {"FPV_DECIDE_PARTS": "2"}), (320, 200, 6, {"FPV_DECIDE_PARTS": "8"}), (320, 200, 6, {"FPV_DECIDE_PARTS": "4", "FPV_DECIDE_BIG": "1"}),
    (320, 200, 6, {"FPV_DECIDE_PARTS": "8", "FPV_TILE_GROUPS": "1"}), (1280, 48, 5, {"FPV_DECIDE_PARTS": "4", "FPV_TILE_GROUPS": "1"}),
    (4096, 1032, 3, {}), (4096, 1032, 3, {"FPV_DECIDE_PARTS": "2"}), (4096, 2064, 2, {})])
def test_decide_in_parts(fpv, synth, oracle, xs, ys, n, env):
    """Few large frames: several workgroups of k_tile_decide share a frame's records and the one that arrives last
    decides (by default for more than 64 tiles per frame -- 4096 x 1032: 129 tiles, 2 parts; 4096 x 2064: 258 tiles, 4
    parts; forced on small frames, also with more parts than tiles); twice per Predictor, so that the tickets have
    to be back at zero for the second batch; noise (decisions flip, frames are redone) and smooth frames, with and
    without a delta frame."""
    import os, torch
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        fr = np.concatenate([synth.plasma_frames(n - n // 2, xs, ys, bits=12, seed=5), (synth.noise_frames(n // 2, xs, ys, seed=6) >> 4)])
        _run_case(fpv, oracle, fr, xs, ys, 4, False, fr[0])
        _run_case(fpv, oracle, fr[1:], xs, ys, 4, False, None)
        p = fpv.Predictor(xs, ys, 4, False, max_batch=n)
        p.set_delta_raw(fr[0])
        d = torch.from_numpy(fr.reshape(n, -1).view(np.int16)).cuda()
        first = [t.clone() for t in p.predict(d.view(torch.uint16))]
        for order in (torch.arange(n - 1, -1, -1), torch.arange(n)):
            got = p.predict(d[order.cuda()].contiguous().view(torch.uint16))
            torch.cuda.synchronize()
            for a, b in zip(first, got):
                assert torch.equal(a[order.cuda()], b)
        p.close()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_randomised_geometries_and_contents(fpv, oracle):
    """Seeded sweep: random multiple-of-4 geometries, shifts, endianness and
    content classes (noise, smooth, constant, near-constant, zero low plane,
    8-bit data) -- flags of every kind must come out identical to the oracle."""
    rng = np.random.default_rng(20240901)
    seen = set()
    for trial in range(24):
        xs = int(rng.integers(1, 90)) * 4
        ys = int(rng.integers(1, 40)) * 4
        be = bool(rng.integers(0, 2))
        shift = int(rng.choice([0, 0, 4, 8, 2, 6]))
        n = 5
        base = rng.integers(0, 65536, (n, ys, xs)).astype(np.uint16)
        kind = trial % 6
        if kind == 1:   # smooth ramps + little noise
            yy, xx = np.mgrid[0:ys, 0:xs]
            base = ((yy * 37 + xx * 53)[None] + rng.integers(0, 64, (n, ys, xs)) + 4000 * np.arange(n)[:, None, None]).astype(np.uint16)
        elif kind == 2:  # constant frames, one with a single different pixel
            base[:] = 0x2345
            base[2, ys // 2, xs // 2] = 0x9abc
            base[3, 0, 0] = 0x1111
        elif kind == 3:  # zero low bytes
            base &= 0xFF00
        elif kind == 4:  # 8-bit data
            base >>= 8
        elif kind == 5:  # frame equals the delta frame
            base[1:] = base[0]
            base[3] = base[0] + 1
        fl = _run_case(fpv, oracle, base, xs, ys, shift, be, base[0])
        seen |= set(int(f) for f in fl)
    assert {0, 1, 3, 4}.issubset(seen) or len(seen) >= 4, seen
