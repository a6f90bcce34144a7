#!/usr/bin/env python3
"""Does the distance between consecutive frames in HBM matter to the three-wave inverse?  All
frames of a batch walk down their rows in step, 1 024 000 (planes) / 2 048 000 (pixels) bytes
apart: every frame's i/o waves touch the same offset modulo 8 KiB at the same time.  Library built
with -DFPV_EXP_PITCH (build/ab/libfpvhip_pitch.so): the same data, planes and pixels at a padded
pitch.   python tools/exp/inv_pitch.py [lib.so] [frames]"""
import ctypes, importlib, os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package
pkg = load_package(); synth = importlib.import_module("fpv_amd.synth")
lib = ctypes.CDLL(os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else "build/ab/libfpvhip_pitch.so"))
pkg._declare_hip(lib); pkg._hip = lib
lib.fpvhip_exp_set_pad.argtypes = [ctypes.c_uint32, ctypes.c_uint32]
XS, YS = 1280, 800
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
NP = XS * YS
base = synth.plasma_frames(64, XS, YS, bits=12, seed=7)
pred = pkg.Predictor(XS, YS, 4, False, max_batch=N); pred.set_delta_raw(base[0])
d = torch.from_numpy(base.reshape(64, -1).view(np.int16)).cuda()
fr = d[torch.arange(N, device="cuda") % 64].contiguous().view(torch.uint16)
lib.fpvhip_exp_set_pad(0, 0)
high, low, pv, fl = pred.predict(fr)
torch.cuda.synchronize()
pred.set_delta_decoded((base[0].astype(np.uint16) << 4).astype(np.uint16))
want = (fr.view(torch.int16).view(N, NP) << 4)
MAXPAD = 8192
ph = torch.zeros(N * (NP + MAXPAD), dtype=torch.uint8, device="cuda")
pl = torch.zeros_like(ph)
rec = torch.zeros(N * (NP + MAXPAD), dtype=torch.int16, device="cuda")
for trial in range(2):
    for pad in [0, 64, 128, 256, 512, 1024, 1024 + 128, 2048, 2048 + 256, 4096, 4096 + 64, 8192, 0]:
        hv = ph[: N * (NP + pad)].view(N, NP + pad); lv = pl[: N * (NP + pad)].view(N, NP + pad)
        hv[:, :NP] = high.view(N, NP); lv[:, :NP] = low.view(N, NP)
        lib.fpvhip_exp_set_pad(pad, pad)
        out = rec.view(torch.uint16)
        pred.inverse(ph[: N * NP], pl[: N * NP], fl, out=out[: N * NP])
        torch.cuda.synchronize()
        got = rec[: N * (NP + pad)].view(N, NP + pad)[:, :NP]
        ok = bool((got == want).all())
        ms = pred.time_inverse(ph[: N * NP], pl[: N * NP], fl, out[: N * NP], 30) / 30
        print(json.dumps({"frames": N, "pad_bytes_planes": pad, "pad_pixels_out": pad, "ms": round(ms, 4), "bit_exact": ok}), flush=True)
