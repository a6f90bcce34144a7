#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: adds the long-stream entries to tests/golden/manifest.json by running
the REAL reference (oracle/_ref/encode = the upstream encode.cc + library compiled here by
oracle/Makefile) on the cycled synthetic frames.  Only data (hashes, sizes) is stored.

  config 1 at its stated size: 1000 frames 1280x800 12-bit plasma, 64 distinct frames cycled.
  config 3 at its stated size: 1000 frames 4096x4096 16-bit plasma, 8 distinct frames cycled
  (`python make_golden_long.py config3`: about a minute of reference CPU time on 8 cores).
  config 2 at its stated size: 10 000 frames 1920x1080 16-bit plasma, 32 distinct frames cycled
  (`python make_golden_long.py config2`: 29.7 GB of stream hashed as it leaves the reference's
  encode.cc; the frame index at its end holds 64-bit offsets beyond 2^32, cc:1185-1197)."""
import hashlib, importlib, json, os, subprocess, sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402
load_package()
synth = importlib.import_module("fpv_amd.synth")


def ref_stream_sha(frames, total, xs, ys, shift, threads=8):
    p = subprocess.Popen([os.path.join(ROOT, "oracle", "_ref", "encode"), str(xs), str(ys), "0", str(shift), str(threads)],
                         stdin=subprocess.PIPE, stdout=subprocess.PIPE)
    h, size = hashlib.sha256(), 0

    import threading
    def feed():
        for i in range(total):
            p.stdin.write(frames[i % len(frames)].tobytes())
        p.stdin.close()
    t = threading.Thread(target=feed); t.start()
    while True:
        b = p.stdout.read(1 << 22)
        if not b:
            break
        h.update(b); size += len(b)
    t.join(); assert p.wait() == 0
    return h.hexdigest(), size


def main():
    man = json.load(open(os.path.join(HERE, "manifest.json")))
    if "config2" in sys.argv[1:]:
        fr = synth.plasma_frames(32, 1920, 1080, bits=16, seed=9)
        sha, size = ref_stream_sha(fr, 10000, 1920, 1080, 0)
        man["workloads"]["plasma16_1920x1080x10000_cycled32_seed9"] = {
            "xs": 1920, "ys": 1080, "shift": 0, "big_endian": False, "frames": 10000, "distinct": 32,
            "note": "frame i = synth.plasma_frames(32, 1920, 1080, bits=16, seed=9)[i % 32]; delta frame = frame 0 (encode.cc:87-90)",
            "fpv_sha256": sha, "fpv_size": size, "generator": "tests/golden/make_golden_long.py config2"}
        json.dump(man, open(os.path.join(HERE, "manifest.json"), "w"), indent=1)
        print(sha, size)
        return
    if "config3" in sys.argv[1:]:
        fr = synth.plasma_frames(8, 4096, 4096, bits=16, seed=11)
        sha, size = ref_stream_sha(fr, 1000, 4096, 4096, 0)
        man["workloads"]["plasma16_4096x4096x1000_cycled8_seed11"] = {
            "xs": 4096, "ys": 4096, "shift": 0, "big_endian": False, "frames": 1000, "distinct": 8,
            "note": "frame i = synth.plasma_frames(8, 4096, 4096, bits=16, seed=11)[i % 8]; delta frame = frame 0 (encode.cc:87-90)",
            "fpv_sha256": sha, "fpv_size": size, "generator": "tests/golden/make_golden_long.py config3"}
        json.dump(man, open(os.path.join(HERE, "manifest.json"), "w"), indent=1)
        print(sha, size)
        return
    fr = synth.plasma_frames(64, 1280, 800, bits=12, seed=7)
    sha, size = ref_stream_sha(fr, 1000, 1280, 800, 4)
    man["workloads"]["plasma12_1280x800x1000_cycled64_seed7"] = {
        "xs": 1280, "ys": 800, "shift": 4, "big_endian": False, "frames": 1000, "distinct": 64,
        "note": "frame i = synth.plasma_frames(64, 1280, 800, bits=12, seed=7)[i % 64]; delta frame = frame 0 (encode.cc:87-90)",
        "fpv_sha256": sha, "fpv_size": size, "generator": "tests/golden/make_golden_long.py"}
    json.dump(man, open(os.path.join(HERE, "manifest.json"), "w"), indent=1)
    print(sha, size)


if __name__ == "__main__":
    main()
