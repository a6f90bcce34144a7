"""bench.py's contract: ONE JSON line on stdout with the fields the driver reads, the roofline
and cpu_baseline objects, and figures that are consistent with each other."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "40", "--warmup", "3",
                        "--no-e2e", "--placement-survey", "1"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 40 and d["warmup"] == 3
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "u8" and d["data"] == "synthetic" and "workload" in d["config"]
    # value = frames of one step / time of one step
    frames = d["config"]["frames_per_step_per_gpu"]
    assert abs(d["value"] - frames / (d["ms_per_step"] / 1e3)) / d["value"] < 0.01
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    # achieved = algorithmic bytes per launch / the dominant kernel's measured duration
    assert rf["kernel"] == "k_tile"                       # 1280x800 takes the tile pass
    k_ms = d["kernel_ms"][rf["kernel"]]
    assert abs(rf["achieved"] - rf["algorithmic_bytes_per_launch"] / (k_ms / 1e3) / 1e9) / rf["achieved"] < 0.01
    assert k_ms <= d["ms_per_step"] * 1.02
    assert 0.3 < rf["frac"] < 1.0
    # traffic is measured in the run itself (rocprofv3 --pmc child processes); the PMC bytes can
    # only exceed the algorithmic ones, and not by much.  Without the profiler on the box the line
    # must still come out, with the reason in it.
    import shutil
    if shutil.which("rocprofv3") is None:
        assert rf["traffic"] is None and "traffic_error" in rf
    else:
        assert "traffic_error" not in rf, rf.get("traffic_error")
        assert rf["traffic"] is not None and "THIS run" in rf["traffic_source"]
        assert 0.98 * rf["algorithmic_bytes_per_launch"] <= rf["traffic"] <= 1.25 * rf["algorithmic_bytes_per_launch"]
        adv = d["adversarial"]
        assert "traffic_error" not in adv and 1.0 < adv["traffic_over_algorithmic"] < 1.8, adv
    # BASELINE config 3's frame size: the order large frames take (groups of frames tile by tile) beats the memory order
    lf = d["large_frames"]
    assert "error" not in lf, lf
    assert lf["flags_histogram"] == {"3": 60} and lf["ms_per_step"] < 0.95 * lf["ms_per_step_frame_by_frame"], lf
    assert lf["inverse"]["roundtrip_bit_exact"] is True and lf["inverse"]["ms_per_batch"] < 0.95 * lf["inverse"]["ms_per_batch_24_pixel_segments"], lf["inverse"]
    mv = d["value_moving_window"]
    assert mv["frames_redone_per_step"] > 0 and 0.5 * d["value"] < mv["frames_per_s"] <= 1.05 * d["value"], mv
    # `value` comes from the first allocation; the survey is reported, never selected
    pl = d["placement"]
    assert abs(pl["first_ms"] - d["ms_per_step"]) / d["ms_per_step"] < 0.05
    assert len(pl["survey_ms_per_step"]) == 1 and pl["best_ms"] <= pl["first_ms"] + 1e-9
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] in ("reference", "port") and cb["cores"] == 1 and cb["value"] > 0
    assert d["inverse"]["roundtrip_bit_exact"] is True
    bc = d["inverse"]["by_content"]
    assert "error" not in bc and all(bc[k]["roundtrip_bit_exact"] and bc[k]["most_passes_in_one_superstep"] <= 40 for k in ("dark_sensor_sigma4_12bit", "smooth_sines_16bit")), bc
    # the power block: present whatever the box offers; where rocm-smi is there, each loop has its samples, and a loaded chip
    # draws more than an idle one
    pw = d["power"]
    assert "error" not in pw and set(("forward", "inverse", "idle_W", "cap_W")) <= set(pw), pw
    if shutil.which("rocm-smi") is not None:
        for k in ("forward", "inverse"):
            assert pw[k]["samples"] >= 2 and pw[k]["socket_W"] > pw["idle_W"] + 100 and 0.3 < pw[k]["ms_per_batch"] < 1.5, pw
            assert pw[k]["ppt_violation_frac"] is None or 0.0 <= pw[k]["ppt_violation_frac"] <= 1.0, pw


def test_two_rank_line_carries_what_the_north_star_asks_for():
    """The multi-GPU line next to the reference CPU encoder on the same host (BASELINE north_star):
    a 2-rank rehearsal on this box's one GPU (gloo control plane, both ranks on GPU 0) must carry
    cpu_baseline, the in-process multi-device Encoder figure (e2e_multi: ONE Encoder over both
    contexts, batches on each) and every rank's kernel times."""
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29617", os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "10", "--warmup", "2", "--share-gpu", "--dist-backend", "gloo",
                        "--placement-survey", "0"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and cb["value"] > 0
    em = d["e2e_multi"]
    assert "error" not in em, em
    assert em["fps"] > 0 and em["devices"] == [0, 0]
    assert len(em["stage_split"]["per_device_batches"]) == 2 and min(em["stage_split"]["per_device_batches"]) > 0
    pr = d["per_rank"]
    assert len(pr["ms_per_step"]) == 2 and len(pr["kernel_ms"]) == 2 and pr["process_group"]["world_size"] == 2
    assert all(k["k_tile"] > 0 for k in pr["kernel_ms"])
    assert abs(d["value"] - 2 * d["config"]["frames_per_step_per_gpu"] / (d["ms_per_step"] / 1e3)) / d["value"] < 0.01


def test_forward_rate_does_not_depend_on_where_the_buffers_lie(fpv, synth):
    """Round 2's headline was measured on the best of sixteen buffer placements (its streaming
    kernel moved by up to 10 % with the location of the buffers in HBM) -- something no product
    context ever did.  bench.py now measures the first allocation; this test pins why that is
    fair: two contexts (each with its own workspace), three independently allocated buffer sets
    each, same 1000-frame batch -- the tile pass's time per batch stays within a few percent."""
    import numpy as np
    import torch
    xs, ys, n = 1280, 800, 1000
    base = synth.plasma_frames(64, xs, ys, bits=12, seed=7)
    d = torch.from_numpy(base.reshape(64, -1).view(np.int16)).cuda()
    idx = torch.arange(n, device="cuda") % 64
    times = []
    keep = []   # hold every set until the end: each lies somewhere else
    for _ in range(2):
        p = fpv.Predictor(xs, ys, 4, False, max_batch=n)
        p.set_delta_raw(base[0])
        for _ in range(3):
            fr = d[idx].contiguous().view(torch.uint16)
            out = p.alloc_outputs(n)
            for _ in range(3):
                p.predict(fr, out=out)
            torch.cuda.synchronize()
            times.append(min(p.time_predict(fr, out, 20) / 20 for _ in range(2)))
            keep.append((fr, out))
        p.close()
    print("ms per 1000-frame batch over six placements:", [round(t, 4) for t in times])
    assert max(times) / min(times) < 1.06, times
