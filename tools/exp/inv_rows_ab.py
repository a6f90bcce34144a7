"""Inverse at widths whose rows are 2 ... 4 supersteps long: the register-north form (round 5: m + 1 register sets, the segment
length that divides the row) against the ring form (FPV_TRIO_RES=0), one process, rounds alternating, round trip checked.
python tools/exp/inv_rows_ab.py"""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package
ENVS = ({}, {"FPV_TRIO_WIDE": "0"}, {"FPV_TRIO_RES": "0"})
pkg = load_package(); synth = importlib.import_module("fpv_amd.synth")
for xs, ys, n, bits, shift in ((4096, 4096, 60, 16, 0), (4096, 4096, 8, 16, 0), (2048, 1536, 256, 12, 4), (2048, 1536, 512, 12, 4), (2048, 2048, 32, 16, 0), (2560, 2048, 128, 12, 4), (2560, 2048, 512, 12, 4), (3072, 3072, 64, 16, 0), (5120, 2880, 16, 16, 0), (1280, 800, 256, 12, 4)):
    base = synth.plasma_frames(9, xs, ys, bits=bits, seed=3)
    d = torch.from_numpy(base[1:].reshape(8, -1).view(np.int16)).cuda()
    fr = d[torch.arange(n, device="cuda") % 8].contiguous().view(torch.uint16)
    p = pkg.Predictor(xs, ys, shift, False, max_batch=n); p.set_delta_raw(base[0])
    planes = p.predict(fr)
    torch.cuda.synchronize()
    p.set_delta_decoded((base[0].astype(np.uint16) << shift).astype(np.uint16))
    rec = torch.empty((n, xs * ys), dtype=torch.uint16, device="cuda")
    res = {}
    for env in ENVS:
        os.environ.update(env)
        rec.zero_()
        p.inverse_pass_stats(reset=True)
        p.inverse(planes[0], planes[1], planes[3], out=rec)
        torch.cuda.synchronize()
        st = p.inverse_pass_stats()
        ok = bool((rec.view(torch.int16) == (fr.view(torch.int16).view(n, xs * ys) << shift)).all())
        res[str(env)] = [ok, st, []]
        for k in env: os.environ.pop(k)
    for _ in range(5):
        for env in ENVS:
            os.environ.update(env)
            res[str(env)][2].append(p.time_inverse(planes[0], planes[1], planes[3], rec, 5) / 5)
            for k in env: os.environ.pop(k)
    for k, (ok, st, t) in res.items():
        print(f"{xs}x{ys} n={n:4d} {k:24s} bit-exact {ok}  supersteps per frame {st[0] // n}  passes per superstep {st[1] / max(st[0], 1):.3f}  ms per batch min {min(t):.4f} median {np.median(t):.4f}", flush=True)
    p.close()
    del fr, rec, planes, d
    torch.cuda.empty_cache()
