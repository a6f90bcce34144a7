#!/usr/bin/env python3
"""How much of the placement effect comes from the CONTEXT's own allocations (delta planes,
workspace)?  Several contexts, ONE set of frame / output buffers, alternating rounds."""
import importlib, os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package
pkg = load_package(); synth = importlib.import_module("fpv_amd.synth")
XS, YS, N = 1280, 800, 1000
base = synth.plasma_frames(64, XS, YS, bits=12, seed=7)
d = torch.from_numpy(base.reshape(64, -1).view(np.int16)).cuda()
fr = d[torch.arange(N, device="cuda") % 64].contiguous().view(torch.uint16)
nctx = int(sys.argv[1]) if len(sys.argv) > 1 else 8
preds, pads = [], []
for k in range(nctx):
    p = pkg.Predictor(XS, YS, 4, False, max_batch=N); p.set_delta_raw(base[0]); preds.append(p)
    pads.append(torch.empty((k * 37 + 5) << 16, dtype=torch.uint8, device="cuda"))  # shifts the next context's allocations
out = preds[0].alloc_outputs(N)
for rnd in range(3):
    row = []
    for p in preds:
        p.predict(fr, out=out); torch.cuda.synchronize()
        row.append(round(p.time_predict(fr, out, 20) / 20, 4))
    print(json.dumps({"round": rnd, "ms_by_context": row}), flush=True)
