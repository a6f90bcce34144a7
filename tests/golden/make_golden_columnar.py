#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: fixtures for the FORMAT parity of the columnar batch
and Arrow record-batch (f4) adapters, produced by the REAL reference: the upstream
columnar_batch.cc / arrow_encoder.cc compiled over the upstream library by oracle/Makefile and
driven single-threaded by tests/columnar_dump.cc / tests/arrow_dump.cc.  Only data is stored
(tests/golden/columnar.npz, arrow_batches.json)."""
import hashlib, importlib, json, os, subprocess, sys, tempfile
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402
load_package()
synth = importlib.import_module("fpv_amd.synth")

XS, YS, SHIFT, N, BATCH = 320, 200, 4, 4, 6


def frames():
    """Smooth, low-frequency 12-bit images: the upstream Batch reserves the stream space of ONE
    uncompressed plane (and one preview) for the WHOLE batch (cb/columnar_batch.cc:36-37) and
    silently stores an empty stream for whatever no longer fits, so only very compressible
    frames fit several to a batch.  (The native Batch reserves batch_size times that.)  The
    frames themselves are stored in the fixture."""
    y, x = np.mgrid[0:YS, 0:XS].astype(np.float64)
    out = []
    for t in range(N):
        v = 2048 + 1500 * np.sin(x / 170.0 + 0.3 * t) * np.cos(y / 130.0 - 0.2 * t) + 300 * np.sin((x + 2 * y) / 260.0 + t)
        out.append((np.round(v).astype(np.int64) >> 3 << 3).clip(0, 4095).astype(np.uint16))
    fr = np.stack(out)
    fr[2, 50:60, 100:140] += 8     # a little structure of its own in one frame
    return fr


def parse_dump(b):
    """The flat form tests/columnar_dump.cc writes."""
    u64 = lambda o: int.from_bytes(b[o:o + 8], "little")
    n, bs = u64(0), u64(8)
    regions = [u64(16 + 8 * k) for k in range(8)]
    o = 16 + 64
    out = {"n": n, "batch_size": bs, "regions": np.array(regions, np.uint64)}
    out["timestamps"] = np.frombuffer(b, "<i8", n, o); o += 8 * n
    out["flags"] = np.frombuffer(b, np.uint8, n, o); o += n
    offs = []
    for c in range(3):
        offs.append(np.frombuffer(b, "<u4", n + 1, o)); o += 4 * (n + 1)
    out["offsets"] = np.stack(offs)
    for c, name in enumerate(("preview", "high", "low")):
        k = int(offs[c][n]); out[name] = np.frombuffer(b, np.uint8, k, o); o += k
    for name in ("delta_high", "delta_low"):
        k = u64(o); o += 8
        out[name] = np.frombuffer(b, np.uint8, k, o); o += k
    assert o == len(b), (o, len(b))
    return out


def main():
    fr = frames()
    with tempfile.TemporaryDirectory() as d:
        raw = os.path.join(d, "frames.raw")
        fr.tofile(raw)
        b = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "columnar_dump"), raw, str(XS), str(YS), str(SHIFT), str(N), str(BATCH)],
                           capture_output=True, check=True).stdout
        dump = parse_dump(b)
        np.savez_compressed(os.path.join(HERE, "columnar.npz"), frames=fr, **{k: np.asarray(v) for k, v in dump.items()})
        print("columnar.npz:", {k: (np.asarray(v).shape if hasattr(v, "shape") else v) for k, v in dump.items()})
        ad = os.path.join(ROOT, "oracle", "_ref", "arrow_dump")
        if os.path.exists(ad):
            # the upstream ArrowEncoder starts its thread before the members the thread uses are
            # constructed: a run can abort or hang; bounded retries
            out = None
            for attempt in range(10):
                try:
                    r = subprocess.run([ad, raw, str(XS), str(YS), str(SHIFT), str(N), "3"], capture_output=True, timeout=20)
                except subprocess.TimeoutExpired:
                    continue
                if r.returncode == 0:
                    out = r.stdout
                    break
            assert out, "the reference Arrow encoder did not complete in 10 attempts"
            open(os.path.join(HERE, "arrow_reference.arrows"), "wb").write(out)
            import pyarrow as pa
            rd = pa.ipc.open_stream(out)
            print("arrow_reference.arrows:", len(out), "bytes;", rd.schema.names, [b.num_rows for b in rd])


if __name__ == "__main__":
    main()
