"""Drop-in proof under pytest: the UPSTREAM encode.cc / decode.cc / benchmark.cc, compiled
unchanged against include/fusion_power_video.h + libfpv_amd.so (build/bin/ref_*, built where
/root/reference exists and shipped with the tree), run on the GPU; compared with the CPU
reference binaries (oracle/_ref, the checker) on the same input and the same host."""
import hashlib
import json
import os
import re
import subprocess
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "build", "bin")
REF = os.path.join(ROOT, "oracle", "_ref")


def _have(*paths):
    return all(os.path.exists(p) for p in paths)


@pytest.fixture(scope="module")
def raw100(synth):
    with tempfile.TemporaryDirectory(dir="/tmp") as d:
        p = os.path.join(d, "plasma100.raw")
        synth.plasma_frames(100, 1280, 800, bits=12).tofile(p)
        yield p


def _run(cmd, stdin_path=None, timeout=120):
    with open(stdin_path, "rb") if stdin_path else open(os.devnull, "rb") as f:
        return subprocess.run(cmd, stdin=f, capture_output=True, timeout=timeout)


@pytest.mark.skipif(not _have(os.path.join(BIN, "ref_encode"), os.path.join(BIN, "ref_decode")),
                    reason="upstream CLIs over this library were not built (no /root/reference at build time)")
def test_upstream_encode_decode_over_the_library(raw100):
    args = ["1280", "800", "0", "4"]   # xsize ysize big_endian shift (enc:41-44)
    shas = {}
    for t in ("0", "8"):
        r = _run([os.path.join(BIN, "ref_encode")] + args + [t], raw100)
        assert r.returncode == 0, r.stderr[-500:]
        shas["upstream encode.cc on the GPU library, threads " + t] = hashlib.sha256(r.stdout).hexdigest()
        stream = r.stdout
    r = _run([os.path.join(BIN, "fpv_encode")] + args + ["16"], raw100)
    assert r.returncode == 0
    shas["own fpv_encode"] = hashlib.sha256(r.stdout).hexdigest()
    if _have(os.path.join(REF, "encode")):
        r = _run([os.path.join(REF, "encode")] + args + ["8"], raw100)
        assert r.returncode == 0
        shas["CPU reference encode"] = hashlib.sha256(r.stdout).hexdigest()
    assert len(set(shas.values())) == 1, shas
    raw = open(raw100, "rb").read()
    with tempfile.NamedTemporaryFile(dir="/tmp", suffix=".fpv") as f:
        f.write(stream)
        f.flush()
        decoders = [[os.path.join(BIN, "ref_decode")] + args, [os.path.join(BIN, "fpv_decode"), "0", "4"]]
        if _have(os.path.join(REF, "decode")):
            decoders.append([os.path.join(REF, "decode")] + args)
        for cmd in decoders:
            r = _run(cmd, f.name)
            assert r.returncode == 0 and r.stdout == raw, cmd


def _bench_ms(binary, raw, threads, reps=3, env=None):
    best = None
    for _ in range(reps):
        r = subprocess.run([binary, raw, "1280", "800", "0", "4", "100", str(threads)], capture_output=True, text=True, timeout=120,
                           env=dict(os.environ, **(env or {})))
        m = re.search(r"^total: .*time: ([0-9.]+) ms", r.stderr, re.M)
        assert r.returncode == 0 and m, r.stderr[-800:]
        assert "roundtrip not equal" not in r.stderr and r.stderr.count("ok") >= 2   # both decoders verified
        best = min(best or 1e9, float(m.group(1)))
    return best


@pytest.mark.skipif(not _have(os.path.join(BIN, "ref_benchmark"), os.path.join(REF, "benchmark")),
                    reason="needs the upstream benchmark built over this library and the CPU reference build")
def test_upstream_benchmark_cold_start_vs_cpu_reference(raw100):
    """benchmark.cc:153-180 times Encoder construction -> Finish of a FRESH process: 100 frames
    1280x800, this library under the unchanged upstream benchmark against the CPU reference on the
    same box, best of three each, in the DEFAULT environment (lazy device bring-up: nothing touches
    the GPU before the first Encoder) and with FPV_WARMUP=eager (bring-up at library load).

    What the default can reach is bounded from below by the HIP runtime itself: tools/exp/cold_probe
    (build/bin/cold_probe) measures what ANY fresh process pays before its first byte comes back from
    the GPU -- runtime init, first launch (code object), a stream, the first transfer each way --
    about 100-120 ms on this pool, i.e. MORE than the CPU reference needs for the whole 100-frame
    job at 32 threads (about 90 ms).  The assertions therefore are: (a) the default adds no more
    than that floor (+35 %) to the eager run -- the library's own share of the cold start is gone --
    (b) regression guards on the default's ratio to the reference, (c) round 2's bounds for eager."""
    floor = None
    probe = os.path.join(BIN, "cold_probe")
    if _have(probe):
        vals = []
        for _ in range(3):
            r = subprocess.run([probe], capture_output=True, text=True, timeout=60)
            m = re.search(r"total ([0-9.]+) ms", r.stdout)
            if r.returncode == 0 and m:
                vals.append(float(m.group(1)))
        floor = min(vals) if vals else None
    table = {"runtime_floor_ms_cold_probe": floor}
    for t in (8, 32, 64):
        ours = _bench_ms(os.path.join(BIN, "ref_benchmark"), raw100, t, env={"FPV_WARMUP": "eager"})
        lazy = _bench_ms(os.path.join(BIN, "ref_benchmark"), raw100, t)
        ref = _bench_ms(os.path.join(REF, "benchmark"), raw100, t)
        table[t] = {"gpu_library_default_ms": lazy, "gpu_library_eager_warmup_ms": ours, "cpu_reference_ms": ref,
                    "fps_ratio_default": round(ref / lazy, 3), "fps_ratio_eager_warmup": round(ref / ours, 3),
                    "default_minus_eager_ms": round(lazy - ours, 1)}
    print(json.dumps(table))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(table, open(os.path.join(ROOT, "gpurun_out", "upstream_benchmark_vs_reference.json"), "w"), indent=1)
    # Timing on a shared box swings (both sides burst beyond the CPU quota for a job this short).
    if floor:   # (64 threads on a 16-CPU quota swing by tens of ms: the bound holds for the calmest of the three, a looser one for all)
        assert min(table[t]["default_minus_eager_ms"] for t in (8, 32, 64)) <= 1.35 * floor + 15.0, table
        for t in (8, 32, 64):
            assert table[t]["default_minus_eager_ms"] <= 2.5 * floor + 40.0, table
    # (the ratio of the default to the reference is the runtime floor over the reference's whole job
    # and swings with the box -- 100 to 160 ms of bring-up were seen on this pool --, so it is
    # recorded, and only guarded against a collapse)
    for t in (8, 32, 64):
        assert table[t]["fps_ratio_default"] >= 0.2, table
    assert table[8]["fps_ratio_eager_warmup"] >= 0.9, table
    assert table[32]["fps_ratio_eager_warmup"] >= 0.9, table
    assert table[64]["fps_ratio_eager_warmup"] >= 0.7, table


@pytest.mark.skipif(not _have(os.path.join(BIN, "frame_rate")), reason="build/bin/frame_rate not built")
def test_frame_api_throughput_next_to_the_reference():
    """The public Frame API (h:75-139) driven like upstream's columnar / Arrow encoders drive it:
    a pool of threads, one `Frame(pixels)` + `Predict(delta)` (or `Compress`) per frame
    (tests/frame_rate.cc, the same source over this library and over the CPU reference).
    The Predict calls of the threads are combined into GPU batches; the plane bytes produced
    are the reference's, and the rate is not below the reference's on the same host."""
    def run(binary, frames, threads):
        r = subprocess.run([binary, "1280", "800", "4", str(frames), str(threads)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-500:]
        return {d["mode"]: d for d in map(json.loads, r.stdout.strip().splitlines())}
    out = {}
    for t in (1, 8, 16):
        run(os.path.join(BIN, "frame_rate"), 64, t)  # warm: contexts, pinned staging
        out[f"native_{t}"] = run(os.path.join(BIN, "frame_rate"), 400, t)
        if _have(os.path.join(REF, "frame_rate")):
            out[f"reference_{t}"] = run(os.path.join(REF, "frame_rate"), 200, t)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "frame_api_rate.json"), "w") as f:
        json.dump(out, f, indent=1)
    for t in (1, 8, 16):
        if f"reference_{t}" not in out:
            continue
        for mode in ("ctor+Predict", "ctor+Compress"):
            n, r = out[f"native_{t}"][mode], out[f"reference_{t}"][mode]
            # same bytes per frame out of both (the compressed sizes in the second mode)
            assert n["plane_bytes"] * r["frames"] == r["plane_bytes"] * n["frames"], (t, mode)
            assert n["fps"] >= 0.9 * r["fps"], (t, mode, n["fps"], r["fps"])
