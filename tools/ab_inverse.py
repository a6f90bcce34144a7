#!/usr/bin/env python3
"""A/B timing of the inverse predictor on one box, ONE process (see tools/ab_forward.py): every
library gets its own context on the SAME planes (made once by the first library's forward
path), timing rounds alternate.  Checks the round trip bit for bit for every library.
  python tools/ab_inverse.py [--frames N] [--rounds R] [--iters I] lib.so[@ENV=VAL,...] ..."""
import argparse, ctypes, importlib, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=1000)
ap.add_argument("--rounds", type=int, default=10)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("libs", nargs="*")
args = ap.parse_args()
pkg = load_package(); synth = importlib.import_module("fpv_amd.synth")
XS, YS, N, SHIFT = 1280, 800, args.frames, 4
NP = XS * YS
base = synth.plasma_frames(64, XS, YS, bits=12, seed=7)
libs = args.libs or [os.path.join(pkg.LIBDIR, "libfpvhip.so")]
d = torch.from_numpy(base.reshape(64, -1).view(np.int16)).cuda()
fr = d[torch.arange(N, device="cuda") % 64].contiguous().view(torch.uint16)
planes = None
rec = torch.empty((N, NP), dtype=torch.uint16, device="cuda")
preds = []
for spec in libs:
    path, _, envs = spec.partition("@")
    added = [kv.split("=", 1) for kv in envs.split(",") if kv]
    for k, v in added:
        os.environ[k] = v
    lib = ctypes.CDLL(os.path.abspath(path)); pkg._declare_hip(lib); pkg._hip = lib
    p = pkg.Predictor(XS, YS, SHIFT, False, max_batch=N)
    p.set_delta_raw(base[0])
    if planes is None:
        planes = p.predict(fr)
        torch.cuda.synchronize()
    p.set_delta_decoded((base[0].astype(np.uint16) << SHIFT).astype(np.uint16))
    rec.zero_()
    p.inverse_pass_stats(reset=True)
    p.inverse(planes[0], planes[1], planes[3], out=rec)
    torch.cuda.synchronize()
    st = p.inverse_pass_stats()
    ok = bool((rec.view(torch.int16) == (fr.view(torch.int16).view(N, NP) << SHIFT)).all())
    preds.append((spec, p, [], ok, st, dict(added)))
    for k, _ in added:
        os.environ.pop(k, None)
for _ in range(args.rounds):
    for spec, p, times, ok, st, env in preds:
        os.environ.update(env)
        times.append(p.time_inverse(planes[0], planes[1], planes[3], rec, args.iters) / args.iters)
        for k in env:
            os.environ.pop(k, None)
ref = None
for spec, p, times, ok, st, env in preds:
    t = np.array(times); med = float(np.median(t)); ref = ref or med
    print(json.dumps({"lib": os.path.basename(spec), "frames": N, "ms_min": round(float(t.min()), 4), "ms_median": round(med, 4),
                      "vs_first": round(med / ref, 4), "frames_per_s": round(N / med * 1e3), "bit_exact": ok,
                      "passes_per_superstep": round(st[1] / max(st[0], 1), 3), "extremes": list(p.last_pass_extremes),
                      "frac_of_8TBps": round(N * 4 * NP / (med / 1e3) / 8e12, 4)}), flush=True)
