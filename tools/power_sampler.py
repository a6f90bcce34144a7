#!/usr/bin/env python3
"""Helper of bench.py's `power` block (started BEFORE the bench touches the GPU, so that the bench never forks once it has):
samples `rocm-smi --showpower --showclocks` of one GPU in a loop and answers commands on stdin, one JSON line each:
  samples   -> {"samples": [[t_mid, sclk_MHz, socket_W], ...]}   (everything since the start)
  throttle  -> {"t": ..., "ppt_acc": ..., "counter": ..., "ppt_activity": ...}   (`amd-smi metric --throttle`: the firmware's
               accumulators of 1 ms ticks -- all ticks, and ticks spent in a package-power (PPT) violation)
  cap       -> {"cap_W": ...}   (`rocm-smi --showmaxpower`)
  on / off  -> {"sampling": true / false}   (the rocm-smi loop; on at the start)
  quit
Standard library only; never touches HIP.  usage: power_sampler.py <gpu index>"""
import json, re, subprocess, sys, threading, time

gpu = sys.argv[1] if len(sys.argv) > 1 else "0"
samples, lock, stop, enabled = [], threading.Lock(), False, True


def run(cmd, timeout=8):
    try:
        return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout).stdout
    except Exception:
        return ""


def sample_loop():
    while not stop:
        if not enabled:
            time.sleep(0.05)
            continue
        t0 = time.time()
        out = run(["rocm-smi", "-d", gpu, "--showpower", "--showclocks", "--csv"])
        t1 = time.time()
        lines = [l for l in out.splitlines() if l.strip()]
        hdr = next((l for l in lines if l.startswith("device")), None)
        row = next((l for l in lines if l.startswith("card")), None)
        if hdr and row:
            h, r = hdr.split(","), row.split(",")
            try:
                ci = next(i for i, x in enumerate(h) if "sclk clock speed" in x)
                pi = next(i for i, x in enumerate(h) if "Power (W)" in x)
                with lock:
                    samples.append([round((t0 + t1) / 2, 3), int(re.sub(r"\D", "", r[ci]) or 0), float(r[pi])])
            except (StopIteration, ValueError, IndexError):
                pass
        time.sleep(0.05)


def find(o, key):
    if isinstance(o, dict):
        for k, v in o.items():
            if k == key:
                return v
            r = find(v, key)
            if r is not None:
                return r
    elif isinstance(o, list):
        for v in o:
            r = find(v, key)
            if r is not None:
                return r
    return None


def throttle():
    out = run(["amd-smi", "metric", "-g", gpu, "--throttle", "--json"], timeout=15)
    try:
        j = json.loads(out[out.index("{"):out.rindex("}") + 1])
        act = find(j, "ppt_violation_activity")

        def xcd_mean(key):   # per-XCD accumulators ({"xcp_0": [8 values]}): their mean
            v = find(j, key)
            v = v.get("xcp_0") if isinstance(v, dict) else v
            v = [x for x in (v or []) if isinstance(x, (int, float))]
            return sum(v) / len(v) if v else None
        return {"t": time.time(), "ppt_acc": int(find(j, "ppt_accumulated")), "counter": int(find(j, "accumulation_counter")),
                "ppt_activity": act.get("value") if isinstance(act, dict) else act,
                "clk_below_limit_acc": xcd_mean("total_gfx_clk_below_host_limit_accumulated"),
                "clk_below_limit_power_acc": xcd_mean("gfx_clk_below_host_limit_power_accumulated"),
                "clk_below_limit_thermal_acc": xcd_mean("gfx_clk_below_host_limit_thermal_accumulated"),
                "low_utilization_acc": xcd_mean("low_utilization_accumulated"),
                "socket_thermal_acc": find(j, "socket_thermal_accumulated"), "hbm_thermal_acc": find(j, "hbm_thermal_accumulated"),
                "vr_thermal_acc": find(j, "vr_thermal_accumulated"), "prochot_acc": find(j, "prochot_accumulated")}
    except Exception:
        return {"t": time.time(), "ppt_acc": None, "counter": None, "ppt_activity": None}


th = threading.Thread(target=sample_loop, daemon=True)
th.start()
for line in sys.stdin:
    cmd = line.strip()
    if cmd == "samples":
        with lock:
            print(json.dumps({"samples": list(samples)}), flush=True)
    elif cmd == "throttle":
        print(json.dumps(throttle()), flush=True)
    elif cmd == "cap":
        m = re.search(r"Max Graphics Package Power \(W\): ([\d.]+)", run(["rocm-smi", "-d", gpu, "--showmaxpower"]))
        print(json.dumps({"cap_W": float(m.group(1)) if m else None}), flush=True)
    elif cmd in ("on", "off"):   # the sampling loop (on at the start): off while the bench times its headline
        enabled = cmd == "on"
        print(json.dumps({"sampling": enabled}), flush=True)
    elif cmd == "quit":
        break
stop = True
