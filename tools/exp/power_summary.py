"""Per phase of tools/exp/power_probe.py: socket power and shader clock (rocm-smi samples) and the PPT throttle accumulator /
violation activity (amd-smi metric --throttle) from the logs tools/exp/power_probe.sh leaves in gpurun_out/.
python tools/exp/power_summary.py [dir]"""
import json, re, sys
d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
phases = []
for l in open(f"{d}/power_probe.log"):
    m = re.match(r"(.{28}) start ([\d.]+) end ([\d.]+)  ms per batch: median ([\d.]+)", l)
    if m: phases.append((m.group(1).strip(), float(m.group(2)), float(m.group(3)), float(m.group(4))))
samples, t = [], None
for l in open(f"{d}/power_samples.log"):
    if l.startswith("t "): t = float(l[2:])
    elif l.startswith("card0") and t:
        f = l.strip().split(",")
        samples.append((t, int(re.sub(r"\D", "", f[5])), float(f[-1])))
parts = re.split(r"^t ([\d.]+)\n", open(f"{d}/power_throttle.log").read(), flags=re.M)
thr = []
for i in range(1, len(parts), 2):
    try:
        j = json.loads(parts[i + 1][parts[i + 1].index("{"):parts[i + 1].rindex("}") + 1])["gpu_data"][0]["throttle"]
    except Exception:
        continue
    thr.append((float(parts[i]), j["ppt_accumulated"], j["accumulation_counter"], j["ppt_violation_activity"]["value"],
                j["socket_thermal_accumulated"], j["hbm_thermal_accumulated"], j["prochot_accumulated"]))
idle = [w for t, c, w in samples if t < phases[0][1] - 0.5 and c < 200]
print(f"idle: {sum(idle) / max(len(idle), 1):.0f} W (sclk ~95 MHz)")
print(f"{'phase':28s} {'ms/batch':>9s} {'W':>6s} {'sclk MHz':>9s} {'PPT activity % (samples)':>26s} {'PPT acc / counter':>18s} {'mJ per frame above idle':>24s}")
for name, a, b, ms in phases:
    s = [(c, w) for t, c, w in samples if a + 0.8 < t < b - 0.2]
    th = [x for x in thr if a - 0.2 < x[0] < b + 1.3]
    w = sum(x[1] for x in s) / len(s); c = sum(x[0] for x in s) / len(s)
    acc = f"{th[-1][1] - th[0][1]} / {th[-1][2] - th[0][2]}" if len(th) > 1 else "-"
    m = re.search(r"(\d+) frames", name)
    e = f"{(w - sum(idle) / max(len(idle), 1)) * ms / int(m.group(1)):.3f}" if m else "-"
    print(f"{name:28s} {ms:9.4f} {w:6.0f} {c:9.0f} {str([x[3] for x in th if a < x[0] < b]):>26s} {acc:>18s} {e:>24s}")
print("thermal / prochot accumulators moved:", any(x[4] != thr[0][4] or x[5] != thr[0][5] or x[6] != thr[0][6] for x in thr))
