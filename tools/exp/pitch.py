#!/usr/bin/env python3
"""Does the distance between consecutive frames in HBM matter?  1280x800 frames are
2 048 000 bytes apart = 500 pages of 4 KiB: every wave of k_stream reads and writes at the
same offset modulo 4 KiB.  Library built with -DFPV_EXP_PITCH (build/ab/libfpvhip_pitch.so):
the same buffers, the same data, frames and planes laid out at a padded pitch.
  python tools/exp/pitch.py [lib.so]"""
import ctypes, importlib, os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package
pkg = load_package(); synth = importlib.import_module("fpv_amd.synth")
lib = ctypes.CDLL(os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else "build/ab/libfpvhip_pitch.so"))
pkg._declare_hip(lib); pkg._hip = lib
lib.fpvhip_exp_set_pad.argtypes = [ctypes.c_uint32, ctypes.c_uint32]
XS, YS, N = 1280, 800, 1000
NP = XS * YS
base = synth.plasma_frames(64, XS, YS, bits=12, seed=7)
pred = pkg.Predictor(XS, YS, 4, False, max_batch=N); pred.set_delta_raw(base[0])
d = torch.from_numpy(base.reshape(64, -1).view(np.int16)).cuda()
idx = torch.arange(N, device="cuda") % 64
MAXPAD = 8192
pads = [0, 64, 128, 256, 512, 1024, 2048, 2048 + 128, 4096, 4096 + 64, 0]
for trial in range(int(os.environ.get("TRIALS", "3"))):
    frames = torch.zeros(N * (NP + MAXPAD), dtype=torch.int16, device="cuda")
    high = torch.empty(N * (NP + MAXPAD), dtype=torch.uint8, device="cuda")
    low = torch.empty_like(high)
    pv = torch.empty((N, NP // 16), dtype=torch.uint8, device="cuda")
    fl = torch.empty(N, dtype=torch.uint8, device="cuda")
    row = {}
    ref = None
    for pad in pads:
        fv = frames[: N * (NP + pad)].view(N, NP + pad)
        fv[:, :NP] = d[idx]
        lib.fpvhip_exp_set_pad(pad, pad)
        out = (high, low, pv, fl)
        f16 = frames.view(torch.uint16)
        for _ in range(3):
            pred.predict(f16[: N * NP], out=out)   # the kernels address by pitch; n comes from numel // npix
        torch.cuda.synchronize()
        ms = pred.time_predict(f16[: N * NP], out, 20) / 20
        h = high[: N * (NP + pad)].view(N, NP + pad)[:, :NP]
        chk = (int(h[3].sum().item()), int(h[999].sum().item()), int(fl.sum().item()))
        ref = ref or chk
        assert chk == ref, (pad, chk, ref)
        row.setdefault(str(pad), []).append(round(ms, 4))
    print(json.dumps({"trial": trial, "ms_by_pad_px": row}), flush=True)
    del frames, high, low
