"""Pass statistics of the speculative inverse on the bench workload (experiments)."""
import sys, os, importlib, numpy as np, torch
sys.path.insert(0, '.')
from __graft_entry__ import load_package
pkg = load_package(); synth = importlib.import_module('fpv_amd.synth')
XS, YS, n = 1280, 800, 64
base = synth.plasma_frames(64, XS, YS, bits=12, seed=7)
pred = pkg.Predictor(XS, YS, 4, False, max_batch=n); pred.set_delta_raw(base[0])
fr = torch.from_numpy(base.reshape(64, -1).view(np.int16)).cuda().view(torch.uint16)
high, low, pv, flags = pred.predict(fr)
pred.set_delta_decoded((base[0].astype(np.uint16) << 4).astype(np.uint16))
out = torch.empty((1, YS, XS), dtype=torch.uint16, device='cuda')
steps = None
for k in range(0, 64, 3):
    pred.inverse_pass_stats(reset=True)
    pred.inverse(high[k:k+1], low[k:k+1], flags[k:k+1], out=out); torch.cuda.synchronize()
    st = pred.inverse_pass_stats(); ex = pred.last_pass_extremes
    ms = pred.time_inverse(high[k:k+1], low[k:k+1], flags[k:k+1], out, 3) / 3
    print(f"frame {k:2d} flags={int(flags[k])} supersteps={st[0]} passes={st[1]} avg={st[1]/max(st[0],1):.2f} max_in_superstep={ex[1]} {ms*1e3:.0f} us", flush=True)
