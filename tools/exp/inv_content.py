"""Passes per superstep and time of the inverse by CONTENT (1280x800, 256 and 1000 frames): plasma, smooth sines + n bits of noise,
uniform noise (shift 0), a photographic-like 1/f field.  python tools/exp/inv_content.py"""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package
pkg = load_package(); synth = importlib.import_module("fpv_amd.synth")
if os.environ.get("FPV_AB_LIB"):   # another build of the library (A/B of a kernel change by content)
    import ctypes
    lib = ctypes.CDLL(os.path.abspath(os.environ["FPV_AB_LIB"])); pkg._declare_hip(lib); pkg._hip = lib
XS, YS = 1280, 800
NP = XS * YS
rng = np.random.default_rng(5)
yy, xx = np.mgrid[0:YS, 0:XS].astype(np.float32)
def sines(k): return 30000 + 9000 * np.sin(xx / (310.0 + 7 * k) + 0.3 * k) + 7000 * np.sin(yy / (270.0 - 5 * k) + 0.2 * k) + 5000 * np.sin((xx + yy) / (530.0 + 11 * k))
def one_over_f(k):
    f = np.fft.rfft2(rng.standard_normal((YS, XS)).astype(np.float32))
    fy, fx = np.meshgrid(np.fft.fftfreq(YS), np.fft.rfftfreq(XS), indexing="ij")
    f /= np.maximum(np.hypot(fy, fx), 1.0 / XS) ** 1.2
    img = np.fft.irfft2(f, s=(YS, XS)); img = (img - img.min()) / (img.max() - img.min())
    return 4000 + 56000 * img
cases = {"plasma 12-bit (the headline's)": (synth.plasma_frames(9, XS, YS, bits=12, seed=7), 4)}
for bits in (0, 2, 4, 6, 8):
    cases[f"sines + {bits} bits of noise, 16-bit"] = (np.stack([np.clip(sines(k) + rng.integers(0, 1 << bits, (YS, XS)), 0, 65535) for k in range(9)]).astype(np.uint16), 0)
cases["1/f field + 4 bits of noise, 16-bit"] = (np.stack([np.clip(one_over_f(k) + rng.integers(0, 16, (YS, XS)), 0, 65535) for k in range(9)]).astype(np.uint16), 0)
cases["uniform noise, 16-bit"] = (synth.noise_frames(9, XS, YS, seed=1), 0)
# a sensor looking at (nearly) nothing: a dark level plus noise, a few bright blobs; the delta frame is another such frame
def dark(sigma, bits):
    blobs = 3000 * np.exp(-((xx - 400) ** 2 + (yy - 300) ** 2) / 5000.0) + 1500 * np.exp(-((xx - 900) ** 2 + (yy - 500) ** 2) / 20000.0)
    return np.stack([np.clip(100 + blobs * (k > 0) + rng.normal(0, sigma, (YS, XS)), 0, (1 << bits) - 1) for k in range(9)]).astype(np.uint16)
for sigma in (1.5, 4, 12):
    cases[f"dark sensor, sigma {sigma}, 12-bit (shift 4)"] = (dark(sigma, 12), 4)
cases["dark sensor, sigma 20, 16-bit (shift 0)"] = (dark(20, 16), 0)
if os.environ.get("CASES"): cases = {k: v for k, v in cases.items() if os.environ["CASES"] in k}
for name, (base, shift) in cases.items():
    for n in ((256,) if os.environ.get("FPV_AB_LIB") else (256, 1000)):
        d = torch.from_numpy(base[1:].reshape(8, -1).view(np.int16)).cuda()
        fr = d[torch.arange(n, device="cuda") % 8].contiguous().view(torch.uint16)
        p = pkg.Predictor(XS, YS, shift, False, max_batch=n); p.set_delta_raw(base[0])
        planes = p.predict(fr)
        torch.cuda.synchronize()
        fl = planes[3].cpu().numpy()
        p.set_delta_decoded((base[0].astype(np.uint16) << shift).astype(np.uint16))
        rec = torch.empty((n, NP), dtype=torch.uint16, device="cuda")
        p.inverse_pass_stats(reset=True)
        p.inverse(planes[0], planes[1], planes[3], out=rec)
        torch.cuda.synchronize()
        st = p.inverse_pass_stats()
        ok = bool((rec.view(torch.int16) == (fr.view(torch.int16).view(n, NP) << shift)).all())
        ms = min(p.time_inverse(planes[0], planes[1], planes[3], rec, 10) / 10 for _ in range(3))
        print(f"{name:40s} n={n:5d} flags {dict(zip(*np.unique(fl, return_counts=True)))} bit-exact {ok} passes per superstep {st[1] / max(st[0], 1):.2f} (max in one {p.last_pass_extremes[1]})  {ms:.4f} ms per batch", flush=True)
        p.close()
        del fr, rec, planes, d
