#!/usr/bin/env python3
"""Time-bounded randomized parity run on the GPU (through the C ABI) against the
CPU oracle (test infrastructure, like everything under tests/): random geometries, shifts, endianness and content classes through
predict and inverse, plus inverse on arbitrary planes with forced flags.

  python tests/fuzz_gpu.py [seconds] [seed]      -> one summary line, exit 1 on a mismatch
  FUZZ_BIG=1: a quarter of the cases use wide / tall frames (up to 8192 x 316)
"""
import importlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402
from oracle.oracle import Oracle          # noqa: E402  (checker only)

pkg = load_package()
orc = Oracle()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)


def content(kind, n, ys, xs):
    if kind == 0:
        return rng.integers(0, 65536, (n, ys, xs)).astype(np.uint16)
    if kind == 1:   # smooth ramps + noise
        yy, xx = np.mgrid[0:ys, 0:xs]
        a, b = rng.integers(1, 90, 2)
        return ((yy * a + xx * b)[None] + rng.integers(0, int(rng.integers(2, 400)), (n, ys, xs))
                + 3000 * np.arange(n)[:, None, None]).astype(np.uint16)
    if kind == 2:   # constant rows (identity chains), a step in the middle
        rows = rng.integers(0, 65536, (n, ys, 1))
        f = np.repeat(rows, xs, axis=2)
        f[:, :, xs // 2:] += rng.integers(0, 5000)
        return f.astype(np.uint16)
    if kind == 3:   # sparse events on a flat background
        f = np.full((n, ys, xs), int(rng.integers(0, 65536)), np.int64)
        m = rng.integers(0, 50, (n, ys, xs)) == 0
        f[m] = rng.integers(0, 65536, int(m.sum()))
        return f.astype(np.uint16)
    if kind == 4:   # zero low bytes / 8-bit data
        f = rng.integers(0, 65536, (n, ys, xs)).astype(np.uint16)
        return (f & 0xFF00) if rng.integers(0, 2) else (f >> 8)
    # blocks of flat values
    by, bx = int(rng.integers(1, 9)), int(rng.integers(1, 40))
    small = rng.integers(0, 65536, (n, (ys + by - 1) // by, (xs + bx - 1) // bx))
    return np.kron(small, np.ones((by, bx), np.int64))[:, :ys, :xs].astype(np.uint16)


cases = frames_total = 0
seen_flags = set()
t0 = t_report = time.time()
while time.time() - t0 < budget:
    if time.time() - t_report > 60:   # a line a minute: long runs must not look hung
        t_report = time.time()
        print(f"... {cases} cases, {frames_total} frames, {t_report - t0:.0f} s", flush=True)
    big = os.environ.get("FUZZ_BIG") and rng.integers(0, 4) == 0     # wide / tall frames now and then
    xs = int(rng.choice([rng.integers(1, 40) * 4, rng.integers(64, 520) * 4, 256, 1280, 1536]))
    ys = int(rng.integers(1, 24)) * 4
    if big:
        xs = int(rng.choice([1920, 2560, 4096, 2048, 3072, 6144, rng.integers(400, 2049) * 4]))   # encode side: xsize <= 8192
        ys = int(rng.integers(8, 80)) * 4
    be = bool(rng.integers(0, 2))
    shift = int(rng.choice([0, 0, 4, 8, 2, 6])) if be else int(rng.choice([0, 0, 4, 8, 2, 6, 12]))
    n = int(rng.integers(1, 6))
    fr = content(int(rng.integers(0, 6)), n, ys, xs)
    if rng.integers(0, 4) == 0:
        fr[1:] = fr[0]          # frames equal to the delta frame
    delta = fr[0] if rng.integers(0, 5) else None
    os.environ["FPV_INVERSE"] = str(rng.choice(["spec", "trio", "auto", "fast"], p=[0.4, 0.4, 0.1, 0.1]))  # read per launch
    # the three-wave kernel: blocks of 1 / 2 / 4 supersteps, residuals in the ring two blocks ahead (round 5) or in their own slots
    os.environ.pop("FPV_TRIO_BLOCK", None)
    if rng.integers(0, 3):    # (unset: the library's own choice, which alone takes the 32-pixel segments)
        os.environ["FPV_TRIO_BLOCK"] = str(rng.choice(["1", "2", "4"]))
    os.environ["FPV_TRIO_INPLACE"] = str(rng.choice(["1", "0"], p=[0.7, 0.3]))
    os.environ.pop("FPV_TRIO_MEMO2", None)
    if rng.integers(0, 2):   # the three-wave kernel's form with the memo behind the second pass (unset: the library's choice from the batches before)
        os.environ["FPV_TRIO_MEMO2"] = str(rng.choice(["1", "0"]))
    os.environ["FPV_TRIO_WIDE"] = str(rng.choice(["1", "0"], p=[0.8, 0.2]))   # 32-pixel segments on rows that are multiples of 2048 pixels (auto kernel choice only)
    os.environ["FPV_TRIO_RES"] = str(rng.choice(["1", "0"], p=[0.7, 0.3]))   # north operands in registers where the row is one superstep long (256, 1280, 1536 wide here)
    # forward path (read when the context is created): the tile pass with a random tile height where the
    # width allows it (multiples of 64), now and then the streaming kernels instead
    for k in ("FPV_TILE_GROUPS", "FPV_FORWARD", "FPV_TILE8_GROUPS", "FPV_TILE8_ROWS", "FPV_TILE_FGROUP", "FPV_DECIDE_PARTS", "FPV_TILE_XBAND"):
        os.environ.pop(k, None)
    if rng.integers(0, 3) == 0:   # k_tile_decide: several workgroups per frame, the last to arrive decides (read per launch)
        os.environ["FPV_DECIDE_PARTS"] = str(rng.choice(["2", "4", "8"]))
    r = int(rng.integers(0, 12))
    if r == 0:
        os.environ["FPV_FORWARD"] = "stream"
    elif r < 4:     # the tile pass (16 pixels per lane) at a random tile height (default height otherwise),
        os.environ["FPV_TILE_GROUPS"] = str(int(rng.integers(1, 24)))
        if rng.integers(0, 2):   # ... in groups of frames walked tile by tile (the order large frames take), band-wise per XCD or not
            os.environ["FPV_TILE_FGROUP"] = str(int(rng.integers(0, 4)))
            os.environ["FPV_TILE_XBAND"] = str(int(rng.integers(0, 2)))
    elif r < 8:     # round 4's variant (8 pixels per lane): 4 or 8 rows per thread, random tile height
        os.environ["FPV_FORWARD"] = "tile8"
        os.environ["FPV_TILE8_ROWS"] = str(int(rng.choice([4, 8])))
        if rng.integers(0, 3):
            os.environ["FPV_TILE8_GROUPS"] = str(int(rng.integers(1, 24)))
    p = pkg.Predictor(xs, ys, shift, be, max_batch=n)
    if delta is not None:
        p.set_delta_raw(delta)
    d = torch.from_numpy(fr.reshape(n, -1).view(np.int16)).cuda().view(torch.uint16)
    high, low, pv, flags = p.predict(d, with_delta=delta is not None)
    torch.cuda.synchronize()
    H, L, P, F = (t.cpu().numpy() for t in (high, low, pv, flags))
    tag = f"case {cases}: {xs}x{ys} shift={shift} be={be} n={n} delta={delta is not None}"
    delta16 = None
    if delta is not None:
        dh, dl, _ = orc.split(delta, shift, be)
        delta16 = (dh.astype(np.uint16) << 8) | dl
        p.set_delta_decoded(delta16)
    for k in range(n):
        eh, el, ep, ef = orc.predict(fr[k], xs, ys, shift, be, delta)
        ok = F[k] == ef and np.array_equal(H[k], eh) and np.array_equal(P[k], ep) and (shift == 8 or np.array_equal(L[k], el))
        if not ok:
            print("PREDICT MISMATCH", tag, "frame", k, flush=True)
            sys.exit(1)
        seen_flags.add(int(ef))
    # a second and third batch of the same frames: now under the flags prediction the first ones left
    # behind (per batch position); the planes must not change
    for _ in range(2):
        h2, l2, p2, f2 = p.predict(d, with_delta=delta is not None)
        torch.cuda.synchronize()
        if not (torch.equal(h2, high) and torch.equal(p2, pv) and torch.equal(f2, flags) and (shift == 8 or torch.equal(l2, low))):
            print("REPEATED BATCH MISMATCH", tag, flush=True)
            sys.exit(1)
    rec = p.inverse(high, low, flags)
    # inverse on arbitrary planes with every flag combination that is legal here
    rh = torch.from_numpy(rng.integers(0, 256, H.shape, dtype=np.uint8)).cuda()
    ff = rng.integers(0, 8, n).astype(np.uint8)
    if delta is None:
        ff &= 6
    rec2 = p.inverse(rh, low, torch.from_numpy(ff).cuda())
    torch.cuda.synchronize()
    rec = rec.cpu().view(torch.int16).numpy().view(np.uint16)
    rec2 = rec2.cpu().view(torch.int16).numpy().view(np.uint16)
    rhn = rh.cpu().numpy()
    for k in range(n):
        e1 = orc.inverse(H[k], None if F[k] & 4 else L[k], delta16, xs, ys, int(F[k]))
        e2 = orc.inverse(rhn[k], None if ff[k] & 4 else L[k], delta16, xs, ys, int(ff[k]))
        if not (np.array_equal(rec[k].reshape(-1), np.asarray(e1).reshape(-1))
                and np.array_equal(rec2[k].reshape(-1), np.asarray(e2).reshape(-1))):
            print("INVERSE MISMATCH", tag, "frame", k, "flags", F[k], ff[k], flush=True)
            sys.exit(1)
    p.close()
    cases += 1
    frames_total += n
print(f"fuzz ok: {cases} cases, {frames_total} frames, flags seen {sorted(seen_flags)}, seed {seed}, {time.time() - t0:.0f} s", flush=True)
