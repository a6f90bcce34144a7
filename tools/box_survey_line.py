#!/usr/bin/env python3
"""One line of profiles/r0N_box_survey.json from a bench.py JSON line on stdin (a few fields of it):
  python bench.py --gpus 1 --steps 20 --warmup 5 | grep '^{' | python tools/box_survey_line.py <box number>"""
import json, sys
d = json.loads(sys.stdin.read())
inv, lf, pw = d.get("inverse") or {}, d.get("large_frames") or {}, d.get("power") or {}
sb = inv.get("smaller_batches") or {}
print(json.dumps({
    "box": int(sys.argv[1]) if len(sys.argv) > 1 else 0, "value": d["value"], "ms_per_step": d["ms_per_step"], "k_tile_ms": d["kernel_ms"].get("k_tile"),
    "roofline_frac": d["roofline"]["frac"], "copy_GBps": d["roofline"].get("measured_copy_GBps"),
    "moving_window_fps": (d.get("value_moving_window") or {}).get("frames_per_s"),
    "inverse_1000_ms": inv.get("ms_per_batch"), "inverse_frac": inv.get("frac"),
    "inverse_ms_32_256_512_768": [sb.get(k, {}).get("ms_per_batch") for k in ("32", "256", "512", "768")],
    "large_frames_ms": lf.get("ms_per_step"), "large_frames_frac": lf.get("frac_all_kernels"),
    "large_frames_inverse_ms": (lf.get("inverse") or {}).get("ms_per_batch"),
    "power_forward_W_ppt": [(pw.get("forward") or {}).get("socket_W"), (pw.get("forward") or {}).get("ppt_violation_frac")],
    "power_inverse_W_ppt": [(pw.get("inverse") or {}).get("socket_W"), (pw.get("inverse") or {}).get("ppt_violation_frac")],
    "e2e_fps": (d.get("e2e") or {}).get("fps"), "e2e_over_reference": (d.get("e2e") or {}).get("over_reference_encoder_same_threads")}))
