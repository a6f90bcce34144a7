"""Is the tile pass on 4096x4096 frames slowed by the frames' power-of-two distance in memory (32 MiB: the same tile of every frame
on the same channels)?  The same pass on frames a few rows taller or shorter.  python tools/exp/stride_probe.py"""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package
pkg = load_package(); synth = importlib.import_module("fpv_amd.synth")
for env in ({}, {"FPV_TILE_XBAND": "1"}):
    os.environ.update(env)
    for xs, ys, n in ((4096, 4096, 60), (4096, 4104, 60), (4096, 4088, 60), (4096, 4160, 60), (4096, 4096, 32), (4096, 4104, 32), (4096, 2048, 120), (4096, 2056, 120), (2048, 2048, 240), (2048, 2056, 240)):
        base = synth.plasma_frames(9, xs, ys, bits=16, seed=3)
        d = torch.from_numpy(base[1:].reshape(8, -1).view(np.int16)).cuda()
        fr = d[torch.arange(n, device="cuda") % 8].contiguous().view(torch.uint16)
        p = pkg.Predictor(xs, ys, 0, False, max_batch=n); p.set_delta_raw(base[0])
        out = p.alloc_outputs(n)
        for _ in range(3): p.predict(fr, out=out)
        torch.cuda.synchronize()
        ms = min(p.time_predict(fr, out, 20) / 20 for _ in range(4))
        _, km = p.time_predict(fr, out, 10, per_kernel=True)
        print(f"{str(env):26s} {xs}x{ys} n={n:4d} shape {p.forward_tile_shape()} {ms:.4f} ms/batch, k_tile {km[0]:.4f} ms = {n * xs * ys * 4.0625 / km[0] / 1e6:.0f} GB/s", flush=True)
        p.close()
        del fr, out, d
        torch.cuda.empty_cache()
    for k in env: os.environ.pop(k)
