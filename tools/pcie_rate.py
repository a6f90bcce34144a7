#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-staged forward path (no Brotli): frames are
already in pinned staging memory; per batch H2D -> kernels -> D2H on the slot's
stream, `slots` batches in flight.  Prints frames/s and GB/s each way."""
import ctypes, importlib, sys, time
import numpy as np
sys.path.insert(0, '.')
from __graft_entry__ import load_package
pkg = load_package(); synth = importlib.import_module('fpv_amd.synth')
lib = pkg.hip_lib()
XS, YS, SHIFT = 1280, 800, 4
base = synth.plasma_frames(8, XS, YS, bits=12)
import os
cases = [tuple(int(v) for v in c.split("x")) for c in os.environ.get("PCIE_CASES", "16x3,16x6,64x3,128x2").split(",")]
for batch, slots in cases:
    ctx = lib.fpvhip_create(0, XS, YS, SHIFT, 0, batch, slots)
    assert ctx, pkg._err()
    assert lib.fpvhip_set_delta_raw(ctx, base[0].ctypes.data) == 0
    for s in range(slots):
        for k in range(batch):
            ctypes.memmove(lib.fpvhip_stage_in(ctx, s, k), base[k % 8].ctypes.data, XS * YS * 2)
    ptrs = [ctypes.c_void_p() for _ in range(4)]
    def wait(s):
        assert lib.fpvhip_predict_wait(ctx, s, *[ctypes.byref(p) for p in ptrs]) == 0
    nb = max(12, 2048 // batch)
    for s in range(slots):
        assert lib.fpvhip_predict_submit(ctx, s, batch, 1) == 0
    t0 = time.perf_counter()
    for b in range(nb):
        s = b % slots
        wait(s)
        lib.fpvhip_predict_submit(ctx, s, batch, 1)
    for s in range(slots):
        wait((nb + s) % slots)
    dt = time.perf_counter() - t0
    frames = (nb + slots) * batch - slots * batch  # the first `slots` submits were outside the timer
    frames = nb * batch
    print(f"batch {batch:4d} slots {slots}: {frames/dt:9.0f} frames/s  H2D {frames*XS*YS*2/dt/1e9:6.1f} GB/s  D2H {frames*XS*YS*2.0625/dt/1e9:6.1f} GB/s")
    lib.fpvhip_destroy(ctx)
