"""Forward-kernel time at other geometries (experiments): python tools/exp/fwd_sizes.py
For each geometry: round 4's tile pass at its default shape, with 8 rows per thread and at other tile
heights, round 3's tile pass (FPV_FORWARD=tile16) and the streaming kernels (FPV_FORWARD=stream), same
frames, same process.  GB/s = algorithmic bytes (4.0625 P per frame)."""
import sys, os, importlib, numpy as np, torch
sys.path.insert(0, '.')
from __graft_entry__ import load_package
pkg = load_package(); synth = importlib.import_module('fpv_amd.synth')
cases = [(1920, 1080, 480, 0, 16), (4096, 4096, 60, 0, 16), (256, 256, 8000, 0, 16), (1024, 1024, 1000, 4, 12), (2048, 1536, 320, 4, 12), (1280, 800, 1000, 4, 12)]
if len(sys.argv) > 1 and sys.argv[1] == 'order':
    cases = [(4096, 4096, 60, 0, 16), (4096, 4096, 125, 0, 16), (4096, 2160, 113, 0, 16), (3072, 3072, 106, 0, 16), (2560, 2048, 190, 4, 12), (2048, 1536, 320, 4, 12), (1920, 1080, 480, 0, 16)]
for xs, ys, n, shift, bits in cases:
    base = synth.plasma_frames(9, xs, ys, bits=bits, seed=3)
    d = torch.from_numpy(base[1:].reshape(8, -1).view(np.int16)).cuda()
    fr = d[torch.arange(n, device='cuda') % 8].contiguous().view(torch.uint16)
    out = None
    # the default (the tile pass, 16 pixels per lane), round 4's variant (8 pixels per lane; R = 4, R = 8, shorter tiles), the streaming kernels
    envs = ({}, {"FPV_FORWARD": "tile8"}, {"FPV_FORWARD": "tile8", "FPV_TILE8_ROWS": "8"}, {"FPV_FORWARD": "tile8", "FPV_TILE8_GROUPS": "3"},
            {"FPV_FORWARD": "stream"})
    if len(sys.argv) > 1 and sys.argv[1] == "order":   # groups of F frames walked tile by tile (large frames by default) 
        envs = ({}, {"FPV_TILE_FGROUP": "0"}, {"FPV_TILE_FGROUP": "5"}, {"FPV_TILE_FGROUP": "6"}, {"FPV_TILE_FGROUP": "7"}, {"FPV_TILE_FGROUP": "8"},
                {"FPV_FORWARD": "stream"})
    for env in envs:
        os.environ.update(env)
        p = pkg.Predictor(xs, ys, shift, False, max_batch=n); p.set_delta_raw(base[0])
        for k in env: os.environ.pop(k)
        out = out or p.alloc_outputs(n)
        for _ in range(3): p.predict(fr, out=out)
        torch.cuda.synchronize()
        ms = min(p.time_predict(fr, out, 10) / 10 for _ in range(3))
        _, km = p.time_predict(fr, out, 5, per_kernel=True)
        print(f"{xs}x{ys} n={n} {str(env):28s} {p.forward_kernel_names()[0]:9s} shape {p.forward_tile_shape()} {ms:.4f} ms/batch  kernels {[round(k,4) for k in km[:3]]}  {n*xs*ys*4.0625/ms/1e6:.0f} GB/s all kernels, {n*xs*ys*4.0625/km[0]/1e6:.0f} dominant", flush=True)
        p.close()
    del fr, out, d
    torch.cuda.empty_cache()
