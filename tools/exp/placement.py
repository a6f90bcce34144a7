#!/usr/bin/env python3
"""How much does the relative placement of the frame / plane buffers in HBM move the
forward kernel?  One arena, the three big buffers at varying byte offsets inside it."""
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
src = d[torch.arange(N, device="cuda") % 64].contiguous().view(torch.uint8).reshape(-1)
GB = 1 << 30
arena = torch.empty(6 * GB, dtype=torch.uint8, device="cuda")
pv = torch.empty((N, NP // 16), dtype=torch.uint8, device="cuda"); fl = torch.empty((N,), dtype=torch.uint8, device="cuda")
def view(off, nbytes, dtype=torch.uint8):
    return arena[off:off + nbytes].view(dtype)
res = []
def run(tag, o_in, o_hi, o_lo):
    fr = view(o_in, N * NP * 2); fr.copy_(src); fr = fr.view(torch.uint16)
    out = (view(o_hi, N * NP).view(N, NP), view(o_lo, N * NP).view(N, NP), pv, fl)
    for _ in range(2): pred.predict(fr, out=out)
    torch.cuda.synchronize()
    ts = [pred.time_predict(fr, out, 20) / 20 for _ in range(5)]
    print(json.dumps({"case": tag, "in": o_in, "high": o_hi, "low": o_lo, "ms_min": round(min(ts), 4), "ms_med": round(float(np.median(ts)), 4)}), flush=True)
for rep in range(2):
    run("natural", 0, 2 * GB, 3 * GB + 512 * 1024 * 1024 * 0)
for off in (256, 1024, 4096, 16384, 65536, 262144, 1 << 20, 3 << 20, 17 << 20):
    run(f"high+{off}", 0, 2 * GB + off, 3 * GB)
    run(f"low+{off}", 0, 2 * GB, 3 * GB + off)
    run(f"both+{off}", 0, 2 * GB + off, 3 * GB + 2 * off)
run("natural", 0, 2 * GB, 3 * GB)
run("tight", 0, N * NP * 2, N * NP * 3)
