#!/usr/bin/env python3
"""Does the WAY the buffers are allocated decide how fast k_stream runs on them?
Strategies, several trials each, all kept alive, timed in alternating rounds:
  sep      five separate torch tensors (what bench.py does)
  arena    ONE allocation of the exact total, carved at 2 MiB boundaries
  arena1g  ONE allocation rounded up to a multiple of 1 GiB
Run a second time with PYTORCH_HIP_ALLOC_CONF=expandable_segments:True for the
virtual-memory-mapped variant of the caching allocator (2 MiB physical chunks)."""
import importlib, os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package
pkg = load_package(); synth = importlib.import_module("fpv_amd.synth")
XS, YS, N = 1280, 800, 1000
NP = XS * YS; NV = NP // 16
base = synth.plasma_frames(64, XS, YS, bits=12, seed=7)
pred = pkg.Predictor(XS, YS, 4, False, max_batch=N); pred.set_delta_raw(base[0])
d = torch.from_numpy(base.reshape(64, -1).view(np.int16)).cuda()
idx = torch.arange(N, device="cuda") % 64
src = d[idx].contiguous().view(torch.uint16)
M2 = 2 << 20

def up(v, a): return (v + a - 1) // a * a

def carve(total_round):
    sizes = [N * NP * 2, N * NP, N * NP, N * NV, N]
    offs, at = [], 0
    for sz in sizes:
        offs.append(at); at = up(at + sz, M2)
    total = up(at, total_round)
    arena = torch.empty(total, dtype=torch.uint8, device="cuda")
    v = [arena[o:o + sz] for o, sz in zip(offs, sizes)]
    fr = v[0].view(torch.uint16); fr.copy_(src.view(-1))
    return arena, fr, (v[1].view(N, NP), v[2].view(N, NP), v[3].view(N, NV), v[4])

sets = []
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for t in range(trials):
    fr = src.clone(); sets.append(("sep", fr, pred.alloc_outputs(N), None))
    a, fr, out = carve(M2); sets.append(("arena", fr, out, a))
    a, fr, out = carve(1 << 30); sets.append(("arena1g", fr, out, a))
for rnd in range(3):
    row = []
    for kind, fr, out, _ in sets:
        pred.predict(fr, out=out); torch.cuda.synchronize()
        row.append((kind, round(pred.time_predict(fr, out, 20) / 20, 4)))
    by = {}
    for k, v in row: by.setdefault(k, []).append(v)
    if rnd == 2:  # a plain copy on the same buffers (first 1e9 bytes of the frames -> the high plane)
        import ctypes
        cp = {}
        for kind, fr, out, _ in sets:
            ms = pkg.hip_lib().fpvhip_time_copy_device(ctypes.c_void_p(out[0].data_ptr()), ctypes.c_void_p(fr.data_ptr()), 1000000000,
                                                       ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), 10)
            cp.setdefault(kind, []).append(round(2e9 * 10 / (ms / 1e3) / 1e9))
            pred.predict(fr, out=out)  # restore the outputs
        print(json.dumps({"copy_GBps": cp}), flush=True)
    print(json.dumps({"round": rnd, "conf": os.environ.get("PYTORCH_HIP_ALLOC_CONF", ""), "ms": by}), flush=True)
