"""Workload for a PMC pass over the tile pass at BASELINE config 3's frame size: python tools/exp/large_pmc.py [xs ys n]
(three forward batches of n frames + the 1 GiB calibration copy); run by tools/exp/large_pmc.sh under rocprofv3 --pmc."""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package
pkg = load_package(); synth = importlib.import_module("fpv_amd.synth")
xs, ys, n = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (4096, 4096, 60)
base = synth.plasma_frames(9, xs, ys, bits=16, seed=3)
d = torch.from_numpy(base[1:].reshape(8, -1).view(np.int16)).cuda()
fr = d[torch.arange(n, device="cuda") % 8].contiguous().view(torch.uint16)
p = pkg.Predictor(xs, ys, 0, False, max_batch=n); p.set_delta_raw(base[0])
out = p.alloc_outputs(n)
for _ in range(3): p.predict(fr, out=out)
pkg.copy_bandwidth_GBps(1 << 30, 2, device=0, shape=1)
torch.cuda.synchronize()
p.close()
