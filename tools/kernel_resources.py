#!/usr/bin/env python3
"""Register / spill / scratch report of the gfx950 kernels (compiles one .hip with
-Rpass-analysis=kernel-resource-usage; nothing runs).  usage: kernel_resources.py file.hip [name-filter] [extra hipcc flags]"""
import re, subprocess, sys
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
extra = sys.argv[3:]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Iinclude",
       "-Ifusion-power-video_amd/csrc", "-c", src, "-o", "/tmp/kernel_resources.o",
       "-Rpass-analysis=kernel-resource-usage"] + extra
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = {}, None
for ln in out.splitlines():
    m = re.search(r"Function Name: (\S+)", ln)
    if m:
        cur = m.group(1); rows[cur] = {}; continue
    m = re.search(r"remark:\s+(.*?): (\d+) \[", ln)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
for k, v in rows.items():
    d = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
    d = re.sub(r"\(fpvhip.*|\(unsigned.*|\(void.*", "", d).replace("fpvhip::(anonymous namespace)::", "").replace("void ", "")
    if flt in d:
        print(f"{d:44s} VGPR {v.get('VGPRs')} AGPR {v.get('AGPRs')} SGPR {v.get('TotalSGPRs')} sgpr-spill {v.get('SGPRs Spill')} "
              f"vgpr-spill {v.get('VGPRs Spill')} scratch {v.get('ScratchSize [bytes/lane]')} occ {v.get('Occupancy [waves/SIMD]')}")
