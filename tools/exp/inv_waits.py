#!/usr/bin/env python3
"""Where the waves of the three-wave inverse wait (library built with -DFPV_EXP_WAITS: clock64 sums per role,
printed by fpvhip_inverse_pass_stats on stderr).  python tools/exp/inv_waits.py build/ab/libfpvhip_waits.so [frames ...]"""
import ctypes, importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package
pkg = load_package(); synth = importlib.import_module("fpv_amd.synth")
lib = ctypes.CDLL(os.path.abspath(sys.argv[1])); pkg._declare_hip(lib); pkg._hip = lib
XS, YS, SHIFT = 1280, 800, 4
NP = XS * YS
base = synth.plasma_frames(64, XS, YS, bits=12, seed=7)
for n in [int(a) for a in sys.argv[2:]] or [1000]:
    p = pkg.Predictor(XS, YS, SHIFT, False, max_batch=n)
    p.set_delta_raw(base[0])
    d = torch.from_numpy(base.reshape(64, -1).view(np.int16)).cuda()
    fr = d[torch.arange(n, device="cuda") % 64].contiguous().view(torch.uint16)
    planes = p.predict(fr)
    p.set_delta_decoded((base[0].astype(np.uint16) << SHIFT).astype(np.uint16))
    rec = torch.empty((n, NP), dtype=torch.uint16, device="cuda")
    for _ in range(3):
        p.inverse(planes[0], planes[1], planes[3], out=rec)
    torch.cuda.synchronize()
    p.inverse_pass_stats(reset=True)
    ms = p.time_inverse(planes[0], planes[1], planes[3], rec, 10) / 10
    sys.stderr.write(f"frames {n}: {ms:.4f} ms per batch ({p.inverse_kernel_name(n)}); sums over 10 launches:\n")
    sys.stderr.flush()
    p.inverse_pass_stats(reset=True)
    p.close()
    del fr, planes, rec
