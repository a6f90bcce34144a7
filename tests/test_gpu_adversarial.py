"""The adversarial forward workload of bench.py at its real size: config-0 style noise at
1280x800 (the clamped-gradient decision flips from frame to frame, so a large part of a batch is
mis-speculated and goes through the exact redo pass, cc:546-564), a handful of frames against
the oracle; and a batch in which every second frame is flat (the delta decision flips: list A,
k_redo_decide)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _check(fpv, oracle, frames, shift, expect_flags=None):
    import torch
    n, ys, xs = frames.shape
    p = fpv.Predictor(xs, ys, shift, False, max_batch=n)
    p.set_delta_raw(frames[0])
    d = torch.from_numpy(frames.reshape(n, -1).view(np.int16)).cuda().view(torch.uint16)
    ctrs = []
    for rep in range(3):   # later batches start from the predictions the earlier ones left behind
        high, low, pv, flags = p.predict(d)
        torch.cuda.synchronize()
        ctrs.append(p.debug_counters())
        fl = flags.cpu().numpy()
        for k in range(n):
            eh, el, ep, ef = oracle.predict(frames[k], xs, ys, shift, False, frames[0])
            assert int(fl[k]) == ef, (rep, k)
            assert np.array_equal(high[k].cpu().numpy(), eh), (rep, k)
            assert np.array_equal(low[k].cpu().numpy(), el), (rep, k)
            assert np.array_equal(pv[k].cpu().numpy(), ep), (rep, k)
    p.close()
    return fl, ctrs


def test_noise_1280x800_mis_speculation(fpv, synth, oracle):
    fr = synth.noise_frames(12, 1280, 800, seed=1)
    fl, ctrs = _check(fpv, oracle, fr, 0)
    assert len(set(int(f) & 2 for f in fl)) == 2          # both CG decisions occur
    assert ctrs[0][1] > 0                                   # first batch: frames went through the redo pass


def test_alternating_flat_frames_flip_the_delta_decision(fpv, synth, oracle):
    fr = synth.plasma_frames(10, 1280, 800, bits=12, seed=7)
    fr[1::2] = 0x0123                                        # E(hist) == 0: no delta (cc:528-533)
    fl, ctrs = _check(fpv, oracle, fr, 4)
    assert {int(f) & 1 for f in fl} == {0, 1}
    assert ctrs[0][0] > 0                                   # list A: delta decision mis-speculated
    # the prediction is per batch position: once a position's decision has repeated it is predicted
    assert ctrs[2][0] == 0 and ctrs[2][1] == 0
