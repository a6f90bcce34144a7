"""world_size-2 `gloo` test of the multi-rank path: frame batches are sharded
(batch b -> rank b % world), every rank encodes its own batches with no
data-path collective, rank 0 gathers the chunks and writes the frame index.
The result must be the byte-identical single-process stream.  CPU only (the
host codec runs against the oracle-backed test shim)."""
import hashlib
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SHIM = os.path.join(HERE, "shim", "libfpv_amd_cpushim.so")

WORKER = r'''
import importlib, os, sys, pickle, hashlib
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
from conftest import load_package
load_package()
synth = importlib.import_module("fpv_amd.synth"); codec_m = importlib.import_module("fpv_amd.codec")
shard = importlib.import_module("fpv_amd.shard")
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
xs, ys, n, batch = 64, 48, 23, 4
frames = synth.noise_frames(n, xs, ys, seed=13)
frames[5:9] = (frames[5:9] >> 9) + 1000          # some CG-on frames
codec = codec_m.Codec(sys.argv[2])
mine = shard.frames_of_rank(n, batch, world, rank)
_, got = codec.encode_with_callbacks(frames[mine], frames[0], xs, ys, threads=2)
header, chunks = got[0][1], [g[1] for g in got[1:-1]]
assert len(chunks) == len(mine)
gathered = [None] * world
dist.all_gather_object(gathered, (mine, chunks))     # control plane only: sizes + bytes for the index
t = torch.tensor([float(len(chunks))]); dist.all_reduce(t, op=dist.ReduceOp.SUM)
assert int(t.item()) == n
if rank == 0:
    by_frame = {}
    for idx, ch in gathered:
        by_frame.update(dict(zip(idx, ch)))
    stream = shard.assemble_stream(header, [by_frame[i] for i in range(n)])
    single = codec.encode_stream(frames, frames[0], xs, ys, threads=3)
    print("RESULT", hashlib.sha256(stream).hexdigest(), hashlib.sha256(single.tobytes()).hexdigest(), flush=True)
dist.barrier(); dist.destroy_process_group()
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_plan_covers_all_frames_once():
    sys.path.insert(0, os.path.join(HERE))
    from conftest import load_package
    load_package()
    import importlib
    shard = importlib.import_module("fpv_amd.shard")
    for n, b, w in ((23, 4, 2), (1000, 32, 8), (5, 8, 4), (0, 4, 2)):
        seen = sorted(sum((shard.frames_of_rank(n, b, w, r) for r in range(w)), []))
        assert seen == list(range(n))
    assert shard.rank_batches(10, 4, 1) == [1, 5, 9]


def test_two_ranks_gloo_sharded_encode_matches_single_process(tmp_path):
    if not os.path.exists(SHIM):
        pytest.skip("tests/shim/libfpv_amd_cpushim.so not built (make shim)")
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, SHIM], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    line = [l for l in outs[0].splitlines() if l.startswith("RESULT")][0].split()
    assert line[1] == line[2], "sharded stream differs from the single-process stream"
