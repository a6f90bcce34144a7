"""Is the clock under the inverse (and the tile pass) a POWER limit?  Loops each kernel for a few seconds per batch size and
prints wall-clock phase stamps; run beside a sampler of rocm-smi (tools/exp/power_probe.sh), which cannot see kernels but
does see socket power and the shader clock.  python tools/exp/power_probe.py"""
import importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package
pkg = load_package(); synth = importlib.import_module("fpv_amd.synth")
XS, YS, SHIFT = 1280, 800, 4
NP = XS * YS
base = synth.plasma_frames(64, XS, YS, bits=12, seed=7)
d = torch.from_numpy(base.reshape(64, -1).view(np.int16)).cuda()
N = 1000
fr = d[torch.arange(N, device="cuda") % 64].contiguous().view(torch.uint16)
p = pkg.Predictor(XS, YS, SHIFT, False, max_batch=N)
p.set_delta_raw(base[0])
out = p.alloc_outputs(N)
planes = p.predict(fr, out=out)
torch.cuda.synchronize()
p.set_delta_decoded((base[0].astype(np.uint16) << SHIFT).astype(np.uint16))
rec = torch.empty((N, NP), dtype=torch.uint16, device="cuda")
def phase(name, fn, seconds=4.0):
    t0 = time.time(); ms = []
    while time.time() - t0 < seconds:
        ms.append(fn())
    print(f"{name:28s} start {t0:.2f} end {time.time():.2f}  ms per batch: median {np.median(ms):.4f} min {min(ms):.4f}", flush=True)
    time.sleep(1.5)   # idle gap: visible in the sampler's log
time.sleep(1.5)
for n in (32, 256, 512, 768, 1000):
    phase(f"inverse {n} frames", lambda: p.time_inverse(planes[0][:n], planes[1][:n], planes[3][:n], rec[:n], 50) / 50)
phase("forward 1000 frames", lambda: p.time_predict(fr, out, 50) / 50)
a = torch.empty(1 << 30, dtype=torch.uint8, device="cuda"); b = torch.empty_like(a)
def copy():
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): b.copy_(a)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20
phase("torch copy 1 GiB", copy)
p.close()
