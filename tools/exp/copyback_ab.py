"""What the copy-back of the inverse's counters behind every launch costs (fpv_capi.hip: 32 bytes into pinned memory, the host's cue for
the memo form of the three-wave kernel): FPV_TRIO_MEMO2 unset (copy, the library decides) against 0 (no copy), alternating rounds.
python tools/exp/copyback_ab.py"""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package
pkg = load_package(); synth = importlib.import_module("fpv_amd.synth")
XS, YS, SHIFT = 1280, 800, 4
base = synth.plasma_frames(64, XS, YS, bits=12, seed=7)
d = torch.from_numpy(base.reshape(64, -1).view(np.int16)).cuda()
for n in (1, 8, 32, 256):
    fr = d[torch.arange(n, device="cuda") % 64].contiguous().view(torch.uint16)
    p = pkg.Predictor(XS, YS, SHIFT, False, max_batch=n); p.set_delta_raw(base[0])
    planes = p.predict(fr); torch.cuda.synchronize()
    p.set_delta_decoded((base[0].astype(np.uint16) << SHIFT).astype(np.uint16))
    rec = torch.empty((n, XS * YS), dtype=torch.uint16, device="cuda")
    res = {"auto": [], "0": []}
    for _ in range(6):
        for k in res:
            if k == "auto": os.environ.pop("FPV_TRIO_MEMO2", None)
            else: os.environ["FPV_TRIO_MEMO2"] = k
            res[k].append(p.time_inverse(planes[0], planes[1], planes[3], rec, 50) / 50)
    os.environ.pop("FPV_TRIO_MEMO2", None)
    print(n, {k: round(float(np.median(v)), 4) for k, v in res.items()}, flush=True)
    p.close()
