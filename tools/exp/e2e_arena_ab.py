import sys, os, importlib, ctypes, json, numpy as np
sys.path.insert(0, '.')
import bench
pkg = bench.load_package(); synth = importlib.import_module("fpv_amd.synth")
base = synth.plasma_frames(64, 1280, 800, bits=12, seed=7)
threads = int(sys.argv[1]) if len(sys.argv) > 1 else bench.host_threads()
r = bench.e2e_encoder(pkg, base, threads, [0], total=int(sys.argv[2]) if len(sys.argv) > 2 else 1000)
print(json.dumps({"arena": os.environ.get("FPV_BROTLI_ARENA", "1"), "threads": threads, "fps": r["fps"], "cold_fps": r["cold_fps"], "warm_100_fps": r["warm_100_fps"],
                  "brotli_ms_per_frame": round(r["stage_split"]["brotli_ms_all_threads"] / r["frames"], 3), "host_stage_over_wall": r["stage_split"]["host_stage_over_wall"],
                  "gpu_batch_latency_ms": r["stage_split"]["gpu_batch_latency_ms_mean"]}))
