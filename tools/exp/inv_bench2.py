"""Inverse-kernel timing (experiments): python tools/exp/inv_bench2.py [n ...]
Run once per FPV_INVERSE = spec | fast to compare kernels (the choice is per process)."""
import sys, os, importlib, numpy as np, torch
sys.path.insert(0, '.')
from __graft_entry__ import load_package
pkg = load_package(); synth = importlib.import_module('fpv_amd.synth')
XS, YS = 1280, 800
ns = [int(a) for a in sys.argv[1:]] or [1, 64, 1024, 4096]
print("FPV_INVERSE =", os.environ.get("FPV_INVERSE", "trio (default)"))
for n in ns:
    base = synth.plasma_frames(min(n, 8) + 1, XS, YS, bits=12)
    pred = pkg.Predictor(XS, YS, 4, False, max_batch=n); pred.set_delta_raw(base[0])
    src = base[1:] if n > 1 else base[1:2]
    d = torch.from_numpy(src.reshape(src.shape[0], -1).view(np.int16)).cuda()
    fr = d[torch.arange(n, device='cuda') % src.shape[0]].contiguous().view(torch.uint16)
    high, low, pv, flags = pred.predict(fr)
    pred.set_delta_decoded((base[0].astype(np.uint16) << 4).astype(np.uint16))  # the decoded delta frame = left-aligned delta image
    out = torch.empty((n, YS, XS), dtype=torch.uint16, device='cuda')
    pred.inverse_pass_stats(reset=True)
    pred.inverse(high, low, flags, out=out); torch.cuda.synchronize()
    st = pred.inverse_pass_stats()
    ok = bool((out.view(torch.int16) == (fr.view(torch.int16).view(n, YS, XS) << 4)).all())
    ms = pred.time_inverse(high, low, flags, out, 5) / 5
    print(f"plasma n={n} ok={ok} flags={np.unique(flags.cpu().numpy())} passes/superstep={st[1]/max(st[0],1):.2f} "
          f"{ms:.3f} ms/batch = {ms/n*1e3:.1f} us/frame, {n/ms*1e3:.0f} fps, {n*4*XS*YS/ms/1e6:.1f} GB/s", flush=True)
    if n <= 1024:
        rnd = torch.randint(0, 256, high.shape, dtype=torch.uint8, device='cuda')
        f2 = torch.full_like(flags, 2)
        pred.inverse_pass_stats(reset=True)
        pred.inverse(rnd, low, f2, out=out); torch.cuda.synchronize()
        st = pred.inverse_pass_stats()
        ms = pred.time_inverse(rnd, low, f2, out, 3) / 3
        print(f"random n={n} passes/superstep={st[1]/max(st[0],1):.2f} {ms:.3f} ms/batch = {ms/n*1e3:.1f} us/frame", flush=True)
    pred.close()
