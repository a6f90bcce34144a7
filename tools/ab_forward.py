#!/usr/bin/env python3
"""A/B timing of the forward predictor on one box, ONE process: every library named on the
command line (default: the in-tree build) gets its own context on the same 1000-frame
batch, and the timing rounds alternate between them, so that clock / box drift hits all
variants alike.  Prints, per library, min and median ms per batch (all forward kernels,
HIP events on the launch stream) and the per-launch split.
  python tools/ab_forward.py [--kind plasma|noise|mixed] [--rounds R] [--iters I] [--nodelta] lib.so ..."""
import argparse, ctypes, importlib, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package

ap = argparse.ArgumentParser()
ap.add_argument("--kind", default="plasma")
ap.add_argument("--rounds", type=int, default=12)
ap.add_argument("--iters", type=int, default=25)
ap.add_argument("--nodelta", action="store_true")
ap.add_argument("libs", nargs="*")
args = ap.parse_args()
pkg = load_package(); synth = importlib.import_module("fpv_amd.synth")
XS, YS, N = 1280, 800, 1000
if args.kind == "plasma":
    base, shift = synth.plasma_frames(64, XS, YS, bits=12, seed=7), 4
elif args.kind == "noise":      # config-0 style content: the CG decision flips from frame to frame
    base, shift = synth.noise_frames(64, XS, YS, seed=1), 0
else:                            # every other frame is flat (its delta decision flips)
    base, shift = synth.plasma_frames(64, XS, YS, bits=12, seed=7), 4
    base[1::2] = 0x0123
libs = args.libs or [os.path.join(pkg.LIBDIR, "libfpvhip.so")]
d = torch.from_numpy(base.reshape(64, -1).view(np.int16)).cuda()
fr = d[torch.arange(N, device="cuda") % 64].contiguous().view(torch.uint16)
preds = []
shared_out = None
for spec in libs:
    path, _, envs = spec.partition("@")      # lib.so@KEY=VAL,KEY2=VAL2: environment while its context is created
    added = [kv.split("=", 1) for kv in envs.split(",") if kv]
    for k, v in added:
        os.environ[k] = v
    lib = ctypes.CDLL(os.path.abspath(path))
    pkg._declare_hip(lib)
    pkg._hip = lib                      # Predictor binds whatever hip_lib() returns
    p = pkg.Predictor(XS, YS, shift, False, max_batch=N)
    if not args.nodelta:
        p.set_delta_raw(base[0])
    if shared_out is None:
        shared_out = p.alloc_outputs(N)
        if os.environ.get("FPV_AB_ILV"):                   # -DFPV_TILE_ILV=1 builds write both planes into `high`: 2 npix per frame (+ slack)
            shared_out = (torch.empty((2 * N + 1, p.npix), dtype=torch.uint8, device="cuda"),) + shared_out[1:]
    out = shared_out                                       # every variant writes the SAME buffers: HBM
    for _ in range(3):                                     # placement alone moves the time by several percent
        p.predict(fr, out=out)
    torch.cuda.synchronize()
    for k, _ in added:
        os.environ.pop(k, None)
    preds.append((spec, p, out, []))
for _ in range(args.rounds):
    for path, p, out, times in preds:
        times.append(p.time_predict(fr, out, args.iters) / args.iters)
ref = None
for path, p, out, times in preds:
    _, km = p.time_predict(fr, out, 10, per_kernel=True)
    fl = out[3].cpu().numpy()
    t = np.array(times)
    med = float(np.median(t))
    ref = ref or med
    print(json.dumps({"lib": os.path.basename(path), "kind": args.kind + ("/nodelta" if args.nodelta else ""),
                      "ms_min": round(float(t.min()), 4), "ms_median": round(med, 4), "vs_first": round(med / ref, 4),
                      "kernel_ms": [round(k, 4) for k in km],
                      "flags": {str(k): int(v) for k, v in zip(*np.unique(fl, return_counts=True))},
                      "frac_of_8TBps_median": round(N * 4160000 / (med / 1e3) / 8e12, 4)}), flush=True)
