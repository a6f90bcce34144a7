"""GPU parity of the speculative segmented inverse kernel (k_inverse_spec, the
replacement of DecompressImage's serial loop cc:326-333) against the oracle.

The kernel guesses each segment's incoming pixel and reruns segments until every
guess is verified, so the interesting inputs are the ones where guesses fail:
random residuals with the CG flag forced on (worst case: up to 64 passes), odd
superstep tails, every segment length, flags of every kind."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["spec", "trio"])
def inverse_kernel(request):
    """Every test of this file runs once per speculative kernel: the one-wave-per-frame
    form (large batches) and the three-wave form (small batches); by default the
    library picks by batch size, which would leave one of them untested here."""
    old = os.environ.get("FPV_INVERSE")
    os.environ["FPV_INVERSE"] = request.param
    yield request.param
    if old is None:
        os.environ.pop("FPV_INVERSE", None)
    else:
        os.environ["FPV_INVERSE"] = old


def _inverse_case(fpv, oracle, high, low, flags, xs, ys, delta16=None):
    import torch
    n = high.shape[0]
    p = fpv.Predictor(xs, ys, 0, False, max_batch=n)
    if delta16 is not None:
        p.set_delta_decoded(delta16)
    th = torch.from_numpy(high.reshape(n, -1)).cuda()
    tl = torch.from_numpy(low.reshape(n, -1)).cuda()
    tf = torch.from_numpy(np.asarray(flags, dtype=np.uint8)).cuda()
    p.inverse_pass_stats(reset=True)
    rec = p.inverse(th, tl, tf)
    torch.cuda.synchronize()
    stats = p.inverse_pass_stats()
    rec = rec.cpu().view(torch.int16).numpy().view(np.uint16)
    for k in range(n):
        exp = oracle.inverse(high[k].reshape(-1), None if flags[k] & 4 else low[k].reshape(-1),
                             delta16, xs, ys, int(flags[k]))
        assert np.array_equal(rec[k].reshape(-1), np.asarray(exp).reshape(-1)), f"frame {k} (flags {flags[k]}) differs"
    p.close()
    return stats


@pytest.mark.parametrize("xs,ys", [(256, 8), (260, 12), (320, 200), (1280, 16), (1284, 8), (1920, 12), (4096, 4), (2560, 4)])
def test_random_residuals_every_flag(fpv, oracle, xs, ys):
    """Uniform random planes: guesses are almost never right, the pass loop runs
    long; all eight flag values, with a delta frame."""
    rng = np.random.default_rng(xs * 7 + ys)
    n = 8
    high = rng.integers(0, 256, (n, ys, xs), dtype=np.uint8)
    low = rng.integers(0, 256, (n, ys, xs), dtype=np.uint8)
    delta16 = rng.integers(0, 65536, ys * xs).astype(np.uint16)
    flags = np.arange(n, dtype=np.uint8)
    steps, passes = _inverse_case(fpv, oracle, high, low, flags, xs, ys, delta16)
    if (ys * xs) % 4 == 0 and xs % 4 == 0:
        assert steps > 0 and passes >= steps       # the speculative kernel ran (CG frames only)


def test_structured_residuals(fpv, oracle):
    """Residual planes with long runs of zeros (identity steps: a wrong guess is
    never forgotten), steps, and saturating values."""
    xs, ys = 1280, 24
    rng = np.random.default_rng(5)
    n = 6
    high = np.zeros((n, ys, xs), dtype=np.uint8)
    high[0] = 0                                   # all zero: x stays at the literal row
    high[1, 0] = rng.integers(0, 256, xs)         # random first row, zero residuals below
    high[2] = (rng.integers(0, 100, (ys, xs)) == 0) * rng.integers(1, 256, (ys, xs))   # sparse
    high[3] = 255                                 # constant -1 steps: wraps mod 256
    high[4] = rng.integers(0, 2, (ys, xs)) * 128  # 0 / 128
    high[5, :, ::7] = 1
    low = rng.integers(0, 256, (n, ys, xs), dtype=np.uint8)
    _inverse_case(fpv, oracle, high, low, np.full(n, 2, np.uint8), xs, ys)


@pytest.mark.parametrize("k4", [1, 2, 3, 4, 5, 6])
def test_every_segment_length(fpv, synth, oracle, k4):
    """FPV_SPEC_K4 forces the segment length (4*k4 pixels per lane); real
    predictor output (plasma) plus noise, odd tail (npix - head not a multiple of
    the superstep)."""
    import torch
    os.environ["FPV_SPEC_K4"] = str(k4)
    try:
        xs, ys = 1536, 20
        fr = np.concatenate([synth.plasma_frames(3, xs, ys, bits=12, seed=9),
                             synth.noise_frames(2, xs, ys, seed=4) >> 4])
        n = fr.shape[0]
        p = fpv.Predictor(xs, ys, 4, False, max_batch=n)
        p.set_delta_raw(fr[0])
        d = torch.from_numpy(fr.reshape(n, -1).view(np.int16)).cuda().view(torch.uint16)
        high, low, pv, flags = p.predict(d)
        flags = (flags | 2)                      # force the chain on every frame
        dh, dl, _ = oracle.split(fr[0], 4, False)
        delta16 = (dh.astype(np.uint16) << 8) | dl
        p.set_delta_decoded(delta16)
        rec = p.inverse(high, low, flags)
        torch.cuda.synchronize()
        rec = rec.cpu().view(torch.int16).numpy().view(np.uint16)
        h, l, f = high.cpu().numpy(), low.cpu().numpy(), flags.cpu().numpy()
        for k in range(n):
            exp = oracle.inverse(h[k], None if f[k] & 4 else l[k], delta16, xs, ys, int(f[k]))
            assert np.array_equal(rec[k].reshape(-1), np.asarray(exp).reshape(-1)), f"frame {k}"
        p.close()
    finally:
        os.environ.pop("FPV_SPEC_K4", None)


def test_pass_counts_separate_easy_from_hard(fpv, synth, oracle):
    """Diagnostics: real predictor residuals need few passes per superstep, random
    residuals many (but never more than 64)."""
    import torch
    xs, ys = 1280, 64
    fr = synth.plasma_frames(4, xs, ys, bits=12, seed=3)
    p = fpv.Predictor(xs, ys, 4, False, max_batch=4)
    p.set_delta_raw(fr[0])
    d = torch.from_numpy(fr.reshape(4, -1).view(np.int16)).cuda().view(torch.uint16)
    high, low, pv, flags = p.predict(d)
    dh, dl, _ = oracle.split(fr[0], 4, False)
    p.set_delta_decoded((dh.astype(np.uint16) << 8) | dl)
    p.inverse_pass_stats(reset=True)
    p.inverse(high, low, flags)
    steps, passes = p.inverse_pass_stats()
    assert steps > 0 and passes / steps < 16, (steps, passes)
    rnd = torch.randint(0, 256, high.shape, dtype=torch.uint8, device="cuda")
    p.inverse(rnd, low, torch.full_like(flags, 2))
    steps2, passes2 = p.inverse_pass_stats()
    assert steps2 > 0 and passes / steps < passes2 / steps2 <= 64, (steps2, passes2)
    p.close()


def _cg_residuals(x, xs):
    """Forward clamped gradient (cc:566-572) of a flat plane with numpy: residuals
    from ORIGINAL neighbours, the first xs + 1 bytes literal."""
    x = x.astype(np.int32).reshape(-1)
    r = x.copy()
    i = np.arange(xs + 1, x.size)
    n, w, nw = x[i - xs], x[i - 1], x[i - xs - 1]
    g = n + w - nw
    pred = np.maximum(np.minimum(n, w), np.minimum(np.maximum(n, w), g))
    r[i] = (x[i] - pred) & 0xff
    return r.astype(np.uint8)


def test_adversarial_flat_rows_cascade(fpv, oracle):
    """The hardest input for the guesses: every image row constant, a different
    constant per row.  The predictor is the identity along such a row (a wrong
    guess is never forgotten) and the pixel above the west neighbour is always the
    wrong guess, so with the plain rule corrections travel one lane per pass (64
    passes per superstep).  The kernel recognises such identity ("shift") segments
    after three passes and proposes the inputs of whole runs of them with a prefix
    scan over the lanes: a handful of passes, and exact either way."""
    import torch
    xs, ys = 1280, 40
    rng = np.random.default_rng(99)
    rows = rng.integers(0, 256, ys).astype(np.uint8)
    img = np.repeat(rows[:, None], xs, axis=1)
    img2 = img.copy()
    img2[:, xs // 2:] = (img2[:, xs // 2:].astype(np.int32) * 7 + 13).astype(np.uint8)  # a step in every row
    high = np.stack([_cg_residuals(img, xs).reshape(ys, xs), _cg_residuals(img2, xs).reshape(ys, xs)])
    low = np.zeros_like(high)
    p = fpv.Predictor(xs, ys, 0, False, max_batch=2)
    p.inverse_pass_stats(reset=True)
    rec = p.inverse(torch.from_numpy(high).cuda(), torch.from_numpy(low).cuda(),
                    torch.tensor([6, 6], dtype=torch.uint8).cuda())     # CG, no low plane
    torch.cuda.synchronize()
    steps, passes = p.inverse_pass_stats()
    rec = rec.cpu().view(torch.int16).numpy().view(np.uint16)
    assert np.array_equal(rec[0] >> 8, img) and np.array_equal(rec[1] >> 8, img2)
    for k in range(2):
        exp = oracle.inverse(high[k].reshape(-1), None, None, xs, ys, 6)
        assert np.array_equal(rec[k].reshape(-1), np.asarray(exp).reshape(-1))
    assert passes / steps < 10, (steps, passes)
    assert p.last_pass_extremes[1] <= 64
    p.close()


@pytest.mark.parametrize("inplace", ["res", "1", "0"])
@pytest.mark.parametrize("block", ["1", "2", "4"])
@pytest.mark.parametrize("xs,ys,n", [(1280, 44, 40), (1024, 28, 24), (1284, 20, 9), (512, 36, 130), (1920, 12, 9), (3072, 8, 5)])
def test_three_wave_block_lengths(fpv, oracle, block, xs, ys, n, inplace):
    """The three-wave kernel meets once per block of M supersteps (1, 2 or 4; FPV_TRIO_BLOCK caps
    it): frames whose superstep count is no multiple of the block, residual and plane slots
    reused block after block, more frames than CUs (several workgroups per CU: every chain wave
    claims a SIMD of its own), flags of every kind.  (Under the one-wave kernel of this file's
    fixture the knob changes nothing: the same cases are one more parity run.)
    All forms of the kernel: residuals loaded into the ring two blocks ahead (round 5; the default
    where the width is a multiple of the segment length: every width here but 1284) -- with the north
    operands kept in registers where the row is exactly one superstep long and the block length even
    ("res", the default: 1280, 1024 and 512 wide here) or read from the ring (FPV_TRIO_RES=0) -- and,
    with FPV_TRIO_INPLACE=0, residuals in slots of their own one block ahead.  1920 and 3072 wide:
    rows longer than a block of one or two supersteps (the ring then holds the row above instead of
    the block being emitted), segments of 24 pixels."""
    old_ip, old_res = os.environ.get("FPV_TRIO_INPLACE"), os.environ.get("FPV_TRIO_RES")
    os.environ["FPV_TRIO_INPLACE"] = "0" if inplace == "0" else "1"
    os.environ["FPV_TRIO_RES"] = "1" if inplace == "res" else "0"
    rng = np.random.default_rng(xs + ys * 3 + n)
    high = rng.integers(0, 256, (n, ys, xs), dtype=np.uint8)
    high[::3] = (rng.integers(0, 40, (len(high[::3]), ys, xs)) == 0) * 7   # long identity runs too
    low = rng.integers(0, 256, (n, ys, xs), dtype=np.uint8)
    delta16 = rng.integers(0, 65536, ys * xs).astype(np.uint16)
    flags = (np.arange(n) % 8).astype(np.uint8)
    old = os.environ.get("FPV_TRIO_BLOCK")
    os.environ["FPV_TRIO_BLOCK"] = block
    try:
        _inverse_case(fpv, oracle, high, low, flags, xs, ys, delta16)
    finally:
        if old is None:
            os.environ.pop("FPV_TRIO_BLOCK", None)
        else:
            os.environ["FPV_TRIO_BLOCK"] = old
        for k, v in (("FPV_TRIO_INPLACE", old_ip), ("FPV_TRIO_RES", old_res)):
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("res", ["default", "0"])
@pytest.mark.parametrize("wide", ["1", "0"])
@pytest.mark.parametrize("xs,ys,n,k4,k4n", [(2560, 20, 9, 10, 5), (3072, 16, 10, 12, 6), (3840, 12, 7, 0, 0), (5120, 12, 6, 10, 5), (4608, 12, 6, 6, 6),
                                              (3072, 4, 5, 12, 6), (2560, 4, 3, 10, 5), (4096, 20, 12, 8, 0), (2048, 24, 12, 8, 0), (6144, 8, 5, 8, 6),
                                              (4096, 4, 5, 8, 0), (2048, 4, 3, 8, 0), (2048, 8, 300, 8, 0)])
def test_rows_of_several_supersteps(fpv, oracle, inverse_kernel, xs, ys, n, k4, k4n, res, wide):
    """Rows that are 2, 3 or 4 supersteps long: the three-wave kernel keeps the north operands in registers there too
    (m + 1 register sets taking turns; round 5) where the geometry's segment length, or the one below it, divides the
    row (2560 / 5120 wide: 20-pixel segments; 3072 / 4608: 24; 3840 wide divides by both and stays at 24 with the ring
    form: k4 = 0 here, either of the geometry's own lengths), for batches the chain bounds.  Rows that are a multiple of
    2048 pixels (2048, 4096, 6144 wide) run 32-pixel segments (k4 = 8: instantiated for this form only, blocks of two),
    1 or 2 times 2560 pixels 40-pixel segments (k4 = 10), 3072 wide 48 (k4 = 12); FPV_TRIO_WIDE=0: the lengths up to 24
    only (k4n).  300 frames of 2048 x 8: more workgroups than compute units.
    Frames with about as many supersteps as a row has (four rows), superstep counts that are no multiple of m + 1 or of
    the block, flags of every kind, long identity runs; FPV_TRIO_RES=0: the same frames with the north bytes from the
    ring.  The superstep count of the batch tells which segment length ran."""
    old_wide = os.environ.get("FPV_TRIO_WIDE")
    os.environ["FPV_TRIO_WIDE"] = wide
    if wide == "0":
        k4 = k4n
    old = os.environ.get("FPV_TRIO_RES")
    if res == "0":
        os.environ["FPV_TRIO_RES"] = "0"
    else:
        os.environ.pop("FPV_TRIO_RES", None)
    try:
        rng = np.random.default_rng(xs + ys * 5 + n)
        high = rng.integers(0, 256, (n, ys, xs), dtype=np.uint8)
        high[::3] = (rng.integers(0, 40, (len(high[::3]), ys, xs)) == 0) * 7
        low = rng.integers(0, 256, (n, ys, xs), dtype=np.uint8)
        delta16 = rng.integers(0, 65536, ys * xs).astype(np.uint16)
        flags = (np.arange(n) % 8).astype(np.uint8)
        steps, passes = _inverse_case(fpv, oracle, high, low, flags, xs, ys, delta16)
        if inverse_kernel == "trio":
            own = min(6, xs // 256)   # the geometry's own segment length; the in-place form may run one below it (xs % (4 k4) == 0)
            fits = []
            for kk in ((k4,) if res == "default" and k4 else (own, own - 1)):
                per_frame = -(-(xs * ys - min(xs + 4, xs * ys)) // (256 * kk))   # the pixels behind the head (first row + 4), in supersteps
                fits.append((steps == 0) if per_frame == 0 else (steps % per_frame == 0 and 1 <= steps // per_frame <= n))
            assert any(fits), (steps, res, k4, own)
    finally:
        for k, v in (("FPV_TRIO_RES", old), ("FPV_TRIO_WIDE", old_wide)):
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("memo2", ["auto", "1", "0"])
def test_smooth_frames_do_not_cascade(fpv, oracle, inverse_kernel, memo2):
    """Smooth frames with a delta frame: the delta planes hover around the 255 / 0 wrap, where the clamped gradient is a shift on
    either side with a constant of its own -- a wrong guess is never forgotten, and under the plain rule its lane's wrong
    output overwrites the right neighbour's right input, which hands a wrong output on: a glitch travelling a lane per pass
    ahead of its fix, up to 60 passes per superstep (round 5 measured 51-58 on such frames).  The long-correction path's memo
    (a lane proposed the input it had before answers with the output it had then) ends it where the fix meets it: the worst
    superstep stays near the point where that path takes over.  Exact either way.
    The three-wave kernel has a second form with such a memo already behind the second pass (FPV_TRIO_MEMO2=1 / 0: always /
    never); left alone, the library takes it for a batch when more than a sixteenth of the supersteps of the batches
    before went into the long-correction path: here from the second launch on, with fewer passes than the first."""
    import torch
    old_m2 = os.environ.get("FPV_TRIO_MEMO2")
    if memo2 == "auto":
        os.environ.pop("FPV_TRIO_MEMO2", None)
    else:
        os.environ["FPV_TRIO_MEMO2"] = memo2
    try:
        _smooth_case(fpv, inverse_kernel, memo2)
    finally:
        if old_m2 is None:
            os.environ.pop("FPV_TRIO_MEMO2", None)
        else:
            os.environ["FPV_TRIO_MEMO2"] = old_m2


def _smooth_case(fpv, inverse_kernel, memo2):
    import torch
    xs, ys, n = 1280, 96, 6
    yy, xx = np.mgrid[0:ys, 0:xs].astype(np.float32)
    fr = np.stack([30000 + 9000 * np.sin(xx / (310.0 + 7 * k) + 0.3 * k) + 7000 * np.sin(yy / (270.0 - 5 * k) + 0.2 * k)
                   + 5000 * np.sin((xx + yy) / (530.0 + 11 * k)) for k in range(n + 1)]).astype(np.uint16)
    p = fpv.Predictor(xs, ys, 0, False, max_batch=n)
    p.set_delta_raw(fr[0])
    d = torch.from_numpy(fr[1:].reshape(n, -1).view(np.int16)).cuda().view(torch.uint16)
    high, low, pv, flags = p.predict(d)
    assert bool(((flags & 3) == 3).all()), flags.cpu().numpy()      # delta + clamped gradient: the case in question
    p.set_delta_decoded(fr[0])
    p.inverse_pass_stats(reset=True)
    rec = p.inverse(high, low, flags)
    torch.cuda.synchronize()
    steps, passes = p.inverse_pass_stats()
    assert np.array_equal(rec.cpu().view(torch.int16).numpy().view(np.uint16).reshape(n, ys, xs), fr[1:])
    assert steps > 0 and passes / steps < 4.5 and p.last_pass_extremes[1] <= 20, (steps, passes, p.last_pass_extremes)
    # a second and a third launch of the same batch: the counters of the launch before have come back by then
    per_launch = [passes / steps]
    for _ in range(2):
        p.inverse_pass_stats(reset=True)
        rec = p.inverse(high, low, flags)
        torch.cuda.synchronize()
        s2, p2 = p.inverse_pass_stats()
        assert np.array_equal(rec.cpu().view(torch.int16).numpy().view(np.uint16).reshape(n, ys, xs), fr[1:])
        per_launch.append(p2 / s2)
    if inverse_kernel == "trio" and memo2 == "auto":
        assert per_launch[2] < per_launch[0] - 0.1, per_launch      # the memo form took over
    if inverse_kernel == "trio" and memo2 == "0":
        assert abs(per_launch[2] - per_launch[0]) < 1e-9, per_launch
    p.close()


def test_chain_simd_claim_fallback_and_recovery(fpv, oracle):
    """The three-wave kernel's workgroups claim a SIMD that no other frame's chain wave is on, in a
    per-CU word of device memory -- a placement hint only.  With every SIMD marked taken (what a
    kernel killed in mid-flight leaves behind) every workgroup takes the fallback choice (wave 0,
    nothing claimed, nothing released): pixels identical; after the reset the claims work again,
    and a healthy run leaves every word clear (a second all-taken run would otherwise differ)."""
    import torch
    xs, ys, n = 1280, 24, 300     # more workgroups than CUs: several per CU at the same time
    rng = np.random.default_rng(77)
    high = rng.integers(0, 256, (n, ys, xs), dtype=np.uint8)
    low = rng.integers(0, 256, (n, ys, xs), dtype=np.uint8)
    flags = (np.arange(n) % 4 | 2).astype(np.uint8)   # every frame has a chain
    delta16 = rng.integers(0, 65536, ys * xs).astype(np.uint16)
    p = fpv.Predictor(xs, ys, 0, False, max_batch=n)
    p.set_delta_decoded(delta16)
    th, tl, tf = (torch.from_numpy(a.reshape(n, -1) if a.ndim == 3 else a).cuda() for a in (high, low, flags))
    want = [np.asarray(oracle.inverse(high[k].reshape(-1), low[k].reshape(-1), delta16, xs, ys, int(flags[k]))).reshape(-1)
            for k in range(0, n, 7)]

    def run():
        rec = p.inverse(th, tl, tf)
        torch.cuda.synchronize()
        return rec.cpu().view(torch.int16).numpy().view(np.uint16).reshape(n, -1)
    base = run()
    for i, k in enumerate(range(0, n, 7)):
        assert np.array_equal(base[k], want[i]), f"frame {k}"
    try:
        p.set_chain_mask(0xF)                      # every SIMD of every CU "taken"
        assert np.array_equal(run(), base)         # the fallback branch, twice (nothing may have been released wrongly)
        assert np.array_equal(run(), base)
    finally:
        p.set_chain_mask(0)
    assert np.array_equal(run(), base)
    p.close()
