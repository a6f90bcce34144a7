#!/usr/bin/env python3
"""Does the inverse pay for its supersteps starting 4 bytes (planes) / 8 bytes (pixels) behind a 128-byte line?  Timing
experiment: the same batch with the plane and pixel buffers shifted so that superstep 0 (pixel xs + 4) starts ON a line
(planes at +124 bytes, pixels at +120 bytes; 1280x800: every frame then does), against the natural placement.
  python tools/exp/inv_align.py [frames ...]"""
import importlib, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package
pkg = load_package(); synth = importlib.import_module("fpv_amd.synth")
XS, YS, SHIFT = 1280, 800, 4
NP = XS * YS
base = synth.plasma_frames(64, XS, YS, bits=12, seed=7)
for n in [int(a) for a in sys.argv[1:]] or [1000, 512]:
    p = pkg.Predictor(XS, YS, SHIFT, False, max_batch=n)
    p.set_delta_raw(base[0])
    d = torch.from_numpy(base.reshape(64, -1).view(np.int16)).cuda()
    fr = d[torch.arange(n, device="cuda") % 64].contiguous().view(torch.uint16)
    hi, lo, pv, fl = p.predict(fr)
    p.set_delta_decoded((base[0].astype(np.uint16) << SHIFT).astype(np.uint16))
    torch.cuda.synchronize()
    variants = {}
    for name, po, oo in (("natural", 0, 0), ("lines", 124, 60), ("natural again", 0, 0), ("planes only", 124, 0), ("pixels only", 0, 60)):
        hb = torch.empty(n * NP + 256, dtype=torch.uint8, device="cuda"); lb = torch.empty_like(hb)
        ob = torch.empty(n * NP + 128, dtype=torch.uint16, device="cuda")
        h2 = hb[po:po + n * NP].view(n, NP); l2 = lb[po:po + n * NP].view(n, NP)
        h2.copy_(hi); l2.copy_(lo)
        o2 = ob[oo:oo + n * NP].view(n, NP)
        p.inverse(h2, l2, fl, out=o2)
        torch.cuda.synchronize()
        ok = bool((o2.view(torch.int16) == (fr.view(torch.int16).view(n, NP) << SHIFT)).all())
        variants[name] = (h2, l2, o2, ok, [], (hb, lb, ob))
    for _ in range(8):
        for name, (h2, l2, o2, ok, times, keep) in variants.items():
            times.append(p.time_inverse(h2, l2, fl, o2, 20) / 20)
    for name, (h2, l2, o2, ok, times, keep) in variants.items():
        print(json.dumps({"frames": n, "placement": name, "bit_exact": ok, "ms_median": round(float(np.median(times)), 4), "ms_min": round(min(times), 4)}), flush=True)
    p.close()
