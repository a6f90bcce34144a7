"""Inverse-kernel time against batch size (experiments): python tools/exp/inv_sizes.py n [n ...]
One context per size; the kernel is whatever launch_inverse picks (FPV_INVERSE / FPV_SPEC_K4 override)."""
import sys, os, importlib, numpy as np, torch
sys.path.insert(0, '.')
from __graft_entry__ import load_package
pkg = load_package(); synth = importlib.import_module('fpv_amd.synth')
XS, YS = 1280, 800
ns = [int(a) for a in sys.argv[1:]] or [512, 1000, 1100, 2000]
base = synth.plasma_frames(65, XS, YS, bits=12, seed=7)
tag = "FPV_INVERSE=%s K4=%s" % (os.environ.get("FPV_INVERSE", "auto"), os.environ.get("FPV_SPEC_K4", "dflt"))
for n in ns:
    pred = pkg.Predictor(XS, YS, 4, False, max_batch=n); pred.set_delta_raw(base[0])
    d = torch.from_numpy(base[1:].reshape(64, -1).view(np.int16)).cuda()
    fr = d[torch.arange(n, device='cuda') % 64].contiguous().view(torch.uint16)
    high, low, pv, flags = pred.predict(fr)
    pred.set_delta_decoded((base[0].astype(np.uint16) << 4).astype(np.uint16))
    out = torch.empty((n, YS, XS), dtype=torch.uint16, device='cuda')
    pred.inverse(high, low, flags, out=out); torch.cuda.synchronize()
    ok = bool((out.view(torch.int16) == (fr.view(torch.int16).view(n, YS, XS) << 4)).all())
    ms = min(pred.time_inverse(high, low, flags, out, 10) / 10 for _ in range(3))
    print(f"{tag} n={n} kernel={pred.inverse_kernel_name(n)} ok={ok} {ms:.4f} ms/batch {n/ms*1e3:.0f} fps {n*4*XS*YS/ms/1e6:.0f} GB/s frac {n*4*XS*YS/ms/1e6/8000:.3f}", flush=True)
    pred.close(); del fr, high, low, pv, flags, out
    torch.cuda.empty_cache()
