"""Large frames: the tile pass's walk with every XCD on its own band of tiles (FPV_TILE_XBAND=1: a tile's delta bytes fetched once
per band and group instead of once per XCD and group) against the default, at several group sizes; one process, rounds
alternating, outputs compared.  python tools/exp/xband_ab.py [xs ys n]"""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package
pkg = load_package(); synth = importlib.import_module("fpv_amd.synth")
xs, ys, n = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (4096, 4096, 60)
base = synth.plasma_frames(9, xs, ys, bits=16, seed=3)
d = torch.from_numpy(base[1:].reshape(8, -1).view(np.int16)).cuda()
fr = d[torch.arange(n, device="cuda") % 8].contiguous().view(torch.uint16)
envs = ({}, {"FPV_TILE_XBAND": "1"}, {}, {"FPV_TILE_XBAND": "1"})
preds, out, first = [], None, None
for env in envs:
    os.environ.update(env)
    p = pkg.Predictor(xs, ys, 0, False, max_batch=n); p.set_delta_raw(base[0])
    out = out or p.alloc_outputs(n)
    for _ in range(3): p.predict(fr, out=out)
    torch.cuda.synchronize()
    got = [o.clone() for o in out]
    if first is None: first = got
    else: assert all(torch.equal(a, b) for a, b in zip(first, got)), f"outputs differ under {env}"
    for k in env: os.environ.pop(k)
    preds.append((env, p, []))
for _ in range(8):
    for env, p, t in preds:
        os.environ.update(env)
        t.append(p.time_predict(fr, out, 20) / 20)
        for k in env: os.environ.pop(k)
for env, p, t in preds:
    os.environ.update(env)
    _, km = p.time_predict(fr, out, 10, per_kernel=True)
    for k in env: os.environ.pop(k)
    t = np.array(t)
    print(f"{xs}x{ys} n={n} {str(env):60s} shape {p.forward_tile_shape()} min {t.min():.4f} median {np.median(t):.4f} ms  launches {[round(k, 4) for k in km[:3]]}", flush=True)
