"""A/B of the forward launches at several geometries, one process, the libraries alternating:
  python tools/exp/finish_ab.py libA.so libB.so
Per geometry and library: ms per batch (all forward launches, min and median of the rounds) and the per-launch split
(k_tile, k_tile_decide, k_tile_finish).  Written for round 5's change of k_tile_finish's work assignment (items of the whole
batch in equal shares instead of whole frames per workgroup), which matters for FEW LARGE frames."""
import ctypes, importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package
pkg = load_package(); synth = importlib.import_module("fpv_amd.synth")
libs = sys.argv[1:] or [os.path.join(pkg.LIBDIR, "libfpvhip.so")]
cases = [(4096, 4096, 60, 0, 16), (4096, 4096, 125, 0, 16), (4096, 4096, 32, 0, 16), (4096, 4096, 16, 0, 16), (4096, 4096, 8, 0, 16), (4096, 4096, 1, 0, 16), (4096, 2160, 113, 0, 16), (4096, 2160, 8, 0, 16), (1920, 1080, 480, 0, 16), (2048, 1536, 320, 4, 12), (1280, 800, 32, 4, 12), (1280, 800, 1000, 4, 12)]
for xs, ys, n, shift, bits in cases:
    base = synth.plasma_frames(9, xs, ys, bits=bits, seed=3)
    d = torch.from_numpy(base[1:].reshape(8, -1).view(np.int16)).cuda()
    fr = d[torch.arange(n, device="cuda") % 8].contiguous().view(torch.uint16)
    preds, out, first = [], None, None
    for path in libs:
        lib = ctypes.CDLL(os.path.abspath(path)); pkg._declare_hip(lib); pkg._hip = lib
        p = pkg.Predictor(xs, ys, shift, False, max_batch=n); p.set_delta_raw(base[0])
        out = out or p.alloc_outputs(n)
        for _ in range(3): p.predict(fr, out=out)
        torch.cuda.synchronize()
        got = [o.clone() for o in out]
        if first is None: first = got
        else: assert all(torch.equal(a, b) for a, b in zip(first, got)), "outputs differ between the libraries"
        preds.append((path, p, []))
    for _ in range(10):
        for path, p, t in preds: t.append(p.time_predict(fr, out, 20) / 20)
    for path, p, t in preds:
        _, km = p.time_predict(fr, out, 10, per_kernel=True)
        t = np.array(t)
        print(f"{xs}x{ys} n={n:5d} {os.path.basename(path):24s} min {t.min():.4f} median {np.median(t):.4f} ms  launches {[round(k, 4) for k in km]}", flush=True)
        p.close()
    del fr, out, d, first, got
    torch.cuda.empty_cache()
