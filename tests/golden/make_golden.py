#!/usr/bin/env python3
"""Generates the golden fixtures from the REAL reference (oracle/_ref, built
from /root/reference by oracle/Makefile).  Run only in the build container:

    python tests/golden/make_golden.py

Writes
  edge_cases.npz   small inputs + the reference's planes / flags / .fpv bytes
  manifest.json    flags and SHA-256 digests for the larger seeded workloads
                   (inputs are regenerated from fusion-power-video_amd/synth.py)
Fixtures are data only: inputs and the outputs the reference produced for them.
"""
import hashlib
import importlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402
from oracle.oracle import Ref  # noqa: E402

load_package()
synth = importlib.import_module("fpv_amd.synth")
ref = Ref()


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def edge_cases():
    """(name, frames[n,48,64] u16, delta u16, shift, big_endian)."""
    xs, ys = 64, 48
    rng = synth.noise_frames(6, xs, ys, seed=11)
    cases = []
    cases.append(("random16", rng[:3], rng[0], 0, False))
    const = np.full((2, ys, xs), 0x1234, np.uint16)
    cases.append(("constant", const, const[0], 0, False))
    near = const.copy()
    near[1, 0, 15] = 0x5634          # one SAMPLED pixel (index 15) differs: entropy edge
    cases.append(("near_constant", near, near[0], 0, False))
    near2 = const.copy()
    near2[1, 0, 16] = 0x5634         # an unsampled pixel differs
    cases.append(("near_constant_unsampled", near2, near2[0], 0, False))
    zl = (rng[1:3] & 0xFF00).copy()  # all-zero low plane against a delta with low bytes
    cases.append(("zero_low_vs_delta_low", zl, rng[0], 0, False))   # flags 5: reference is lossy here
    zl0 = np.zeros((2, ys, xs), np.uint16)
    cases.append(("all_zero", zl0, zl0[0], 0, False))
    p12 = synth.plasma_frames(3, xs, ys, bits=12, seed=3)
    cases.append(("plasma12_le_s4", p12, p12[0], 4, False))
    cases.append(("plasma12_le_s0", p12, p12[0], 0, False))
    cases.append(("plasma12_be_s4", p12.byteswap(), p12[0].byteswap(), 4, True))
    cases.append(("plasma12_be_s0", p12.byteswap(), p12[0].byteswap(), 0, True))
    p8 = (p12 >> 4).astype(np.uint16)
    cases.append(("plasma8_le_s8", p8, p8[0], 8, False))
    cases.append(("plasma8_be_s8", p8.byteswap(), p8[0].byteswap(), 8, True))
    cases.append(("random16_be", rng[3:6], rng[3], 0, True))
    cases.append(("random16_be_s3", rng[3:6], rng[3], 3, True))
    cases.append(("random16_le_s5", rng[3:6], rng[3], 5, False))
    smooth = (p12 << 4) + (rng[:3] >> 12)      # 16-bit smooth data: CG on
    cases.append(("smooth16", smooth.astype(np.uint16), smooth[0].astype(np.uint16), 0, False))
    return cases


def main():
    out = {}
    names = []
    for name, frames, delta, shift, be in edge_cases():
        n, ys, xs = frames.shape
        names.append(name)
        out[name + "/frames"] = frames
        out[name + "/delta"] = delta
        out[name + "/meta"] = np.array([xs, ys, shift, int(be)], np.int64)
        highs, lows, pvs, flags = [], [], [], []
        for k in range(n):
            h, l, pv, f = ref.predict(frames[k], xs, ys, shift, be, delta)
            highs.append(h); lows.append(l if l.size else np.zeros(0, np.uint8)); pvs.append(pv); flags.append(f)
        out[name + "/high"] = np.stack(highs)
        out[name + "/low"] = np.stack(lows)
        out[name + "/preview"] = np.stack(pvs)
        out[name + "/flags"] = np.array(flags, np.uint8)
        # the delta frame itself, predicted without a delta frame
        h, l, pv, f = ref.predict(delta, xs, ys, shift, be, None)
        out[name + "/delta_high"] = h
        out[name + "/delta_flags"] = np.array([f], np.uint8)
        fpv = ref.encode_stream(frames, delta, xs, ys, shift, be, threads=0)
        out[name + "/fpv"] = fpv
        dec, dpv = ref.decode_stream(fpv, with_previews=True)
        out[name + "/decoded"] = dec
        out[name + "/decoded_preview"] = dpv
        print(f"{name:28s} flags={flags} delta_flags={f} fpv={fpv.size}B "
              f"lossless={np.array_equal(dec, ref_shifted(frames, shift, be))}")
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "edge_cases.npz"), **out)

    man = {"brotli_version": "1.0.9", "generator": "tests/golden/make_golden.py", "workloads": {}}
    # config 0: 256x256 noise x 64
    nz = synth.noise_frames(64, 256, 256, seed=1)
    fl = [ref.predict(nz[k], 256, 256, 0, False, nz[0])[3] for k in range(64)]
    fpv = ref.encode_stream(nz, nz[0], 256, 256, 0, False, threads=4)
    man["workloads"]["noise_256x256x64_seed1"] = {
        "xs": 256, "ys": 256, "shift": 0, "big_endian": False, "frames": 64,
        "input_sha256": sha(nz), "flags": fl, "fpv_sha256": sha(fpv), "fpv_size": int(fpv.size)}
    print("noise flags:", fl)
    # config 1 shape: 1280x800 plasma, 12-bit, shift 4
    pl = synth.plasma_frames(20, 1280, 800, bits=12, seed=7)
    planes = []
    for k in range(3):
        h, l, pv, f = ref.predict(pl[k], 1280, 800, 4, False, pl[0])
        planes.append({"flags": f, "high_sha256": sha(h), "low_sha256": sha(l), "preview_sha256": sha(pv)})
    fpv = ref.encode_stream(pl, pl[0], 1280, 800, 4, False, threads=8)
    man["workloads"]["plasma12_1280x800x20_seed7"] = {
        "xs": 1280, "ys": 800, "shift": 4, "big_endian": False, "frames": 20,
        "input_sha256": sha(pl), "planes": planes, "fpv_sha256": sha(fpv), "fpv_size": int(fpv.size)}
    # config 2 shape: 1920x1080 16-bit
    p16 = synth.plasma_frames(4, 1920, 1080, bits=16, seed=9)
    fpv = ref.encode_stream(p16, p16[0], 1920, 1080, 0, False, threads=4)
    man["workloads"]["plasma16_1920x1080x4_seed9"] = {
        "xs": 1920, "ys": 1080, "shift": 0, "big_endian": False, "frames": 4,
        "input_sha256": sha(p16), "flags": [ref.predict(p16[k], 1920, 1080, 0, False, p16[0])[3] for k in range(4)],
        "fpv_sha256": sha(fpv), "fpv_size": int(fpv.size)}
    # config 3 shape: 4096x4096 16-bit (2 frames)
    p4k = synth.plasma_frames(2, 4096, 4096, bits=16, seed=5)
    fpv = ref.encode_stream(p4k, p4k[0], 4096, 4096, 0, False, threads=2)
    man["workloads"]["plasma16_4096x4096x2_seed5"] = {
        "xs": 4096, "ys": 4096, "shift": 0, "big_endian": False, "frames": 2,
        "input_sha256": sha(p4k), "flags": [ref.predict(p4k[k], 4096, 4096, 0, False, p4k[0])[3] for k in range(2)],
        "fpv_sha256": sha(fpv), "fpv_size": int(fpv.size)}
    json.dump(man, open(os.path.join(HERE, "manifest.json"), "w"), indent=1)
    frame_api_golden()
    print(json.dumps({k: v["fpv_size"] for k, v in man["workloads"].items()}))


def frame_api_golden():
    """Frame public API (h:75-139) through the reference: fixtures for
    tests/test_frame_api.py."""
    import ctypes
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from frame_api_driver import FrameApi, cases
    api = FrameApi(ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libfpvref.so")), "ref_")
    xs, ys, cs = cases(synth)
    out = {}
    for name, img, delta, shift, be in cs:
        out[name + "/full"] = api.compress_full(img, delta, xs, ys, shift, be)
        if delta is not None:
            for k, v in api.cycle(img, delta, xs, ys, shift, be).items():
                out[name + "/" + k] = v
    fr = synth.plasma_frames(2, xs, ys, bits=12, seed=3)
    out["frame8/core"] = api.compress8_core((fr[1] >> 4).astype(np.uint8), (fr[0] >> 4).astype(np.uint8), xs, ys)
    np.savez_compressed(os.path.join(HERE, "frame_api.npz"), **out)
    print("frame_api.npz:", len(out), "arrays")


def ref_shifted(frames, shift, be):
    f = frames.byteswap() if be else frames
    return (f.astype(np.uint32) << shift).astype(np.uint16)


if __name__ == "__main__":
    main()
