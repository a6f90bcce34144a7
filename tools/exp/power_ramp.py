"""How much does the power limit cost?  After two seconds of idle the chip's power average starts low: chunks of 5 launches,
timed one after the other, show the kernel before the limit engages (if it ever runs faster) and in the steady state.
python tools/exp/power_ramp.py   (forward 1000 frames, inverse 1000 / 512 frames, 1280x800)"""
import importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package
pkg = load_package(); synth = importlib.import_module("fpv_amd.synth")
XS, YS, SHIFT, N = 1280, 800, 4, 1000
NP = XS * YS
base = synth.plasma_frames(64, XS, YS, bits=12, seed=7)
d = torch.from_numpy(base.reshape(64, -1).view(np.int16)).cuda()
fr = d[torch.arange(N, device="cuda") % 64].contiguous().view(torch.uint16)
p = pkg.Predictor(XS, YS, SHIFT, False, max_batch=N)
p.set_delta_raw(base[0])
out = p.alloc_outputs(N)
planes = p.predict(fr, out=out)
torch.cuda.synchronize()
p.set_delta_decoded((base[0].astype(np.uint16) << SHIFT).astype(np.uint16))
rec = torch.empty((N, NP), dtype=torch.uint16, device="cuda")
def series(name, fn, chunks=240):
    for rep in range(2):
        time.sleep(2.0)
        t0 = time.time()
        ms = [fn() for _ in range(chunks)]
        wall = time.time() - t0
        a = np.array(ms)
        print(f"{name} (run {rep}): chunks of 5 launches after 2 s of idle, ms per launch; wall {wall:.2f} s")
        print("  first 24 chunks:", " ".join(f"{x:.4f}" for x in a[:24]))
        for k in range(0, chunks, 40):
            print(f"  chunks {k:3d}-{k + 39:3d}: mean {a[k:k + 40].mean():.4f} min {a[k:k + 40].min():.4f} max {a[k:k + 40].max():.4f}")
        sys.stdout.flush()
series("forward 1000 frames", lambda: p.time_predict(fr, out, 5) / 5)
series("inverse 1000 frames", lambda: p.time_inverse(planes[0], planes[1], planes[3], rec, 5) / 5)
series("inverse 512 frames", lambda: p.time_inverse(planes[0][:512], planes[1][:512], planes[3][:512], rec[:512], 5) / 5)
p.close()
