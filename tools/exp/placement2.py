#!/usr/bin/env python3
"""Spread of k_stream's time over independent buffer sets allocated one after another in one
process (all kept alive): is the placement of a set in HBM a property one can choose?"""
import importlib, os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package
pkg = load_package(); synth = importlib.import_module("fpv_amd.synth")
XS, YS, N = 1280, 800, 1000
NP = XS * YS
base = synth.plasma_frames(64, XS, YS, bits=12, seed=7)
pred = pkg.Predictor(XS, YS, 4, False, max_batch=N); pred.set_delta_raw(base[0])
d = torch.from_numpy(base.reshape(64, -1).view(np.int16)).cuda()
idx = torch.arange(N, device="cuda") % 64
sets = []
for t in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    fr = d[idx].contiguous().view(torch.uint16)
    out = pred.alloc_outputs(N)
    sets.append((fr, out))
for rnd in range(3):
    row = []
    for fr, out in sets:
        pred.predict(fr, out=out); torch.cuda.synchronize()
        row.append(round(pred.time_predict(fr, out, 20) / 20, 4))
    print(json.dumps({"round": rnd, "ms": row, "ptr_GB": [round(fr.data_ptr() / 2**30, 2) for fr, _ in sets]}), flush=True)
