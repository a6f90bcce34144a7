#!/usr/bin/env python3
"""Buffer sets allocated one after another; each is timed, then k_stream runs ONCE more on
each, in order: under `rocprofv3 --pmc ...` the last len(sets) k_stream dispatches belong to
the sets in order, so counters can be laid beside the (un-profiled-quality) timings.
  rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum ... -d out -- python3 tools/exp/placement_pmc.py 5"""
import importlib, os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package
pkg = load_package(); synth = importlib.import_module("fpv_amd.synth")
XS, YS, N = 1280, 800, 1000
base = synth.plasma_frames(64, XS, YS, bits=12, seed=7)
pred = pkg.Predictor(XS, YS, 4, False, max_batch=N); pred.set_delta_raw(base[0])
d = torch.from_numpy(base.reshape(64, -1).view(np.int16)).cuda()
src = d[torch.arange(N, device="cuda") % 64].contiguous().view(torch.uint16)
sets = [(src.clone(), pred.alloc_outputs(N)) for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5)]
ms = []
for fr, out in sets:
    pred.predict(fr, out=out); torch.cuda.synchronize()
    ms.append(round(pred.time_predict(fr, out, 10) / 10, 4))
for fr, out in sets:   # the marked dispatches
    pred.predict(fr, out=out); torch.cuda.synchronize()
print(json.dumps({"ms": ms, "frames_ptr_GB": [round(fr.data_ptr() / 2**30, 3) for fr, _ in sets],
                  "high_ptr_GB": [round(o[0].data_ptr() / 2**30, 3) for _, o in sets]}), flush=True)
