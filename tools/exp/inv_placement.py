#!/usr/bin/env python3
"""Where do the chain waves of k_inverse_trio run (library built with -DFPV_EXP_HWID)?  Each
chain wave leaves its HW_ID / XCC_ID and its lifetime in the first pixels of its frame.
  python tools/exp/inv_placement.py build/ab/libinv_hwid.so [frames] [FPV_TRIO_BLOCK]"""
import collections, ctypes, importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package
pkg = load_package(); synth = importlib.import_module("fpv_amd.synth")
XS, YS, SHIFT = 1280, 800, 4
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
if len(sys.argv) > 3: os.environ["FPV_TRIO_BLOCK"] = sys.argv[3]
NP = XS * YS
base = synth.plasma_frames(64, XS, YS, bits=12, seed=7)
lib = ctypes.CDLL(os.path.abspath(sys.argv[1])); pkg._declare_hip(lib); pkg._hip = lib
d = torch.from_numpy(base.reshape(64, -1).view(np.int16)).cuda()
fr = d[torch.arange(N, device="cuda") % 64].contiguous().view(torch.uint16)
p = pkg.Predictor(XS, YS, SHIFT, False, max_batch=N)
p.set_delta_raw(base[0])
planes = p.predict(fr)
p.set_delta_decoded((base[0].astype(np.uint16) << SHIFT).astype(np.uint16))
rec = torch.empty((N, NP), dtype=torch.uint16, device="cuda")
for rep in range(3):
    p.inverse(planes[0], planes[1], planes[3], out=rec)
torch.cuda.synchronize()
flags = planes[3].cpu().numpy()
w = rec[:, :4].cpu().numpy().view(np.uint32).reshape(N, 2)
cg = (flags & 2) != 0
hw, clk = w[cg, 0], w[cg, 1].astype(np.float64) * 16
xcc, simd, cu, se = hw >> 16, (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 13) & 7
per_simd = collections.Counter(zip(xcc.tolist(), se.tolist(), cu.tolist(), simd.tolist()))
share = np.array([per_simd[k] for k in zip(xcc.tolist(), se.tolist(), cu.tolist(), simd.tolist())])
per_cu = collections.Counter(zip(xcc.tolist(), se.tolist(), cu.tolist()))
print("frames with a chain:", int(cg.sum()), " CUs used:", len(per_cu), " chain waves per CU:", dict(collections.Counter(per_cu.values())))
print("chain waves per SIMD (how many SIMDs):", dict(collections.Counter(per_simd.values())))
for k in sorted(set(share.tolist())):
    sel = share == k
    print("chain waves sharing their SIMD with %d other chain wave(s): %4d   lifetime clocks mean %.0f  max %.0f" % (k - 1, sel.sum(), clk[sel].mean(), clk[sel].max()))
print("all: mean %.0f  p50 %.0f  p99 %.0f  max %.0f" % (clk.mean(), np.median(clk), np.percentile(clk, 99), clk.max()))
