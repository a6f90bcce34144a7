"""Inverse kernel with and without the delta planes' reads (experiments): python tools/exp/inv_nodelta.py [frames]
The same 1000 plasma frames encoded with a delta frame (flags 3: the inverse re-reads 2 bytes per pixel of
delta planes from L2 for every frame) and without one (flags 2): what do those L2 hits cost the inverse?"""
import sys, os, importlib, numpy as np, torch
sys.path.insert(0, '.')
from __graft_entry__ import load_package
pkg = load_package(); synth = importlib.import_module('fpv_amd.synth')
XS, YS, N = 1280, 800, int(sys.argv[1]) if len(sys.argv) > 1 else 1000
base = synth.plasma_frames(64, XS, YS, bits=12, seed=7)
d = torch.from_numpy(base.reshape(64, -1).view(np.int16)).cuda()
fr = d[torch.arange(N, device='cuda') % 64].contiguous().view(torch.uint16)
res = {}
ps = []
for tag, use in (("with delta", True), ("no delta", False)):
    p = pkg.Predictor(XS, YS, 4, False, max_batch=N)
    if use:
        p.set_delta_raw(base[0])
    high, low, pv, flags = p.predict(fr, with_delta=use)
    torch.cuda.synchronize()
    ps.append((tag, p, high, low, flags, use))
out = torch.empty((N, XS * YS), dtype=torch.uint16, device='cuda')
for tag, p, high, low, flags, use in ps:
    if use:
        # the decoder's delta frame = the shifted delta pixels
        p.set_delta_decoded(((base[0].astype(np.uint32) << 4) & 0xffff).astype(np.uint16))
    for _ in range(3):
        p.inverse(high, low, flags, out=out)
    torch.cuda.synchronize()
for rnd in range(6):
    for tag, p, high, low, flags, use in ps:
        ms = p.time_inverse(high, low, flags, out, 20) / 20
        res.setdefault(tag, []).append(ms)
for tag, v in res.items():
    fl = dict(zip(*np.unique(ps[0][4].cpu().numpy() if tag == "with delta" else ps[1][4].cpu().numpy(), return_counts=True)))
    print(f"{tag:12s} {N} frames: min {min(v):.4f} median {sorted(v)[len(v)//2]:.4f} ms  flags {fl}  kernel {p.inverse_kernel_name(N) if hasattr(p,'inverse_kernel_name') else ''}")
