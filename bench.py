#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X predictor/residual path.

Workload (BASELINE.json configs[1]): 1280x800, 12-bit-in-16 synthetic plasma,
1000 frames, encode direction.  One "step" = one pass of the forward predictor
(split + preview + delta/CG decisions + residuals: everything Frame's ctor and
Frame::Predict do in the reference) over the 1000-frame batch, inputs resident
in HBM.  Brotli is not part of the hot path (it stays on the host cores).

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.  Multi-GPU: frame batches are sharded, one process
per GPU, no data-path collective (weak scaling: 1000 frames per GPU).
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

XS, YS, SHIFT, NFRAMES, DISTINCT = 1280, 800, 4, 1000, 64
NPIX = XS * YS
ALG_BYTES_PER_FRAME = 2 * NPIX + NPIX + NPIX + NPIX // 16   # 4.0625 * P
HBM_PEAK_GBPS = 8000.0                                      # MI355X_MICROARCH.md: 8 TB/s spec


def cpu_quota():
    """CPUs this container may actually use (cgroup v2 quota), or None."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if q == "max" else round(int(q) / int(per), 2)
    except Exception:
        return None


def cpu_baseline(frames, seconds=12.0):
    """Time the CPU checker on this host's cores over a bounded sample of the
    same workload: Frame ctor + Frame::Predict per frame, single thread."""
    from oracle.oracle import Oracle, Ref
    kind = "reference" if Ref.available() else "port"
    impl = Ref() if kind == "reference" else Oracle()
    delta = frames[0]
    impl.predict(frames[1], XS, YS, SHIFT, False, delta)  # warm
    n, t0 = 0, time.perf_counter()
    while True:
        impl.predict(frames[n % len(frames)], XS, YS, SHIFT, False, delta)
        n += 1
        el = time.perf_counter() - t0
        if el >= seconds:
            break
    return {"value": round(n / el, 2), "unit": "frames/s", "cores": 1, "kind": kind,
            "sample": f"{n} frames of the same 1280x800 plasma workload, Frame ctor + Predict, 1 thread, {el:.1f} s",
            "host_cpus": os.cpu_count(), "host_cpu_quota": cpu_quota()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=NFRAMES)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (RCCL, one GPU per rank) or gloo (rehearsal)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal on a box with fewer GPUs than ranks: rank r uses GPU r %% device_count")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if args.share_gpu:
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        # The ranks never exchange frame data: the process group only carries the
        # barrier and the max-over-ranks of the elapsed time.
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.dist_backend)

    pkg = load_package()
    synth = importlib.import_module("fpv_amd.synth")
    # 64 distinct plasma frames (different phase window per rank), cycled to the batch size
    base = synth.plasma_frames(DISTINCT, XS, YS, bits=12, seed=7, first=rank * DISTINCT)
    n = args.frames
    pred = pkg.Predictor(XS, YS, SHIFT, False, max_batch=n, slots=0, device=local_rank)
    pred.set_delta_raw(base[0])
    d_base = torch.from_numpy(base.reshape(DISTINCT, -1).view(np.int16)).to(dev)
    idx = torch.arange(n, device=dev) % DISTINCT
    d_frames = d_base[idx].contiguous().view(torch.uint16)
    del d_base
    out = pred.alloc_outputs(n)

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        pred.predict(d_frames, out=out)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pred.predict(d_frames, out=out)
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=dev if args.dist_backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # HIP-event timing on the launch stream: whole pipeline and per kernel
    ev_ms, kernel_ms = pred.time_predict(d_frames, out, max(args.steps, 5), per_kernel=True)
    names = ["k_stream", "k_redo_decide", "k_redo_stream", "-"]
    copy_gbps = pkg.copy_bandwidth_GBps(1 << 30, 5, device=local_rank) if rank == 0 else 0.0
    dom = int(np.argmax(kernel_ms))
    dom_s = kernel_ms[dom] / 1e3
    achieved = n * ALG_BYTES_PER_FRAME / dom_s / 1e9
    flags = out[3].cpu().numpy()

    # Decode direction on the same batch (secondary figure, not `value`): the inverse
    # predictor (DecompressImage's pixel loops, cc:326-344) over the planes just made.
    inv = None
    if rank == 0:
        # the decoded delta frame is the left-aligned delta image (cc:437-443 then cc:340-341)
        pred.set_delta_decoded((base[0].astype(np.uint16) << SHIFT).astype(np.uint16))
        rec = torch.empty((n, NPIX), dtype=torch.uint16, device=dev)
        pred.inverse_pass_stats(reset=True)
        pred.inverse(out[0], out[1], out[3], out=rec)
        torch.cuda.synchronize()
        st = pred.inverse_pass_stats()
        exact = bool((rec.view(torch.int16) == (d_frames.view(torch.int16).view(n, NPIX) << SHIFT)).all())
        it = max(args.steps // 2, 3)
        inv_ms = pred.time_inverse(out[0], out[1], out[3], rec, it) / it
        inv_bytes = n * 4 * NPIX                                  # B_dec = 4 * P
        inv = {"kernel": "k_inverse_spec", "frames_per_s": round(n / inv_ms * 1e3, 1),
               "ms_per_batch": round(inv_ms, 4), "bound": "hbm",
               "achieved": round(inv_bytes / inv_ms / 1e6, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
               "frac": round(inv_bytes / inv_ms / 1e6 / HBM_PEAK_GBPS, 4),
               "algorithmic_bytes_per_launch": inv_bytes,
               "passes_per_superstep": round(st[1] / max(st[0], 1), 3),
               "roundtrip_bit_exact": exact}
        del rec

    if rank == 0:
        fps = world * n * args.steps / elapsed
        res = {
            "metric": "encode frames/sec at 1280x800x16-bit (predictor/residual stage, bit-exact vs reference)",
            "value": round(fps, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": "1280x800 12-bit-in-16 synthetic plasma, 1000 frames per GPU per step, shift=4, frame 0 = delta frame",
                       "frames_per_step_per_gpu": n, "parallelism": f"frame batches sharded over {world} GPU(s), no collectives"},
            "input_GBps": round(fps * 2 * NPIX / 1e9, 1),
            "algorithmic_GBps_all_kernels": round(n * ALG_BYTES_PER_FRAME / (ev_ms / max(args.steps, 5) / 1e3) / 1e9, 1),
            "kernel_ms": {k: round(v, 4) for k, v in zip(names, kernel_ms)},
            "flags_histogram": {str(k): int(v) for k, v in zip(*np.unique(flags, return_counts=True))},
            "roofline": {"bound": "hbm", "kernel": names[dom], "achieved": round(achieved, 1),
                         "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
                         "traffic": None,
                         "algorithmic_bytes_per_launch": n * ALG_BYTES_PER_FRAME,
                         "measured_copy_GBps": round(copy_gbps, 1)},
        }
        if inv is not None:
            res["inverse"] = inv
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tf):
            try:
                tj = json.load(open(tf))
                t = tj.get(names[dom])
                if t and n == NFRAMES:   # PMC bytes per launch of the same 1000-frame workload
                    res["roofline"]["traffic"] = t["bytes_per_launch"]
                    res["roofline"]["traffic_source"] = t["source"]
                if inv is not None and n == NFRAMES and tj.get(inv["kernel"]):
                    res["inverse"]["traffic"] = tj[inv["kernel"]]["bytes_per_launch"]
            except Exception:
                pass
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(base)
        print(json.dumps(res), flush=True)
    pred.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
