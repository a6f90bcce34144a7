#!/usr/bin/env python3
"""HBM traffic per kernel launch from rocprofv3 --pmc passes (tools/prof_pmc.sh).

usage: traffic_from_pmc.py gpurun_out/pmc_<tag> > profiles/traffic.json

FETCH_SIZE / WRITE_SIZE are reported in KiB per dispatch.  Corrections, as
MI355X_MICROARCH.md (HBM / rocprofv3) prescribes for gfx950: FETCH_SIZE counts a
wide coalesced streaming read at half its bytes -> doubled; WRITE_SIZE is exact
for 16-B-per-lane streaming stores.  One k_stream dispatch per batch (round 2 on; round 1
launched a pair per batch)."""
import collections
import re
import csv
import glob
import json
import os
import sys

root = sys.argv[1]
tag = os.path.basename(os.path.normpath(root))
per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
            name = r["Kernel_Name"]
            m = re.search(r"\b(k_[a-z0-9_]+)", name)
            key = m.group(1) if m else None
            if key:
                per[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
# read-side factor: calibrated on the 1 GiB copy of the same run when it is there (bench.py --pmc-child),
# else the guide's 2.0 for gfx950
cal = per.get("k_copy_simple", {})
f_rd = (1 << 30) / (1024 * sum(cal["FETCH_SIZE"]) / len(cal["FETCH_SIZE"])) if cal.get("FETCH_SIZE") else 2.0
f_wr = (1 << 30) / (1024 * sum(cal["WRITE_SIZE"]) / len(cal["WRITE_SIZE"])) if cal.get("WRITE_SIZE") else 1.0
for k, cs in per.items():
    if not cs.get("FETCH_SIZE") or not cs.get("WRITE_SIZE"):
        continue
    group = 1                                     # dispatches per logical launch
    rd = f_rd * 1024 * sum(cs["FETCH_SIZE"]) / (len(cs["FETCH_SIZE"]) / group)
    wr = f_wr * 1024 * sum(cs["WRITE_SIZE"]) / (len(cs["WRITE_SIZE"]) / group)
    out[k] = {"bytes_per_launch": int(rd + wr), "read_bytes": int(rd), "write_bytes": int(wr),
              "dispatches_sampled": len(cs["FETCH_SIZE"]), "read_factor": round(f_rd, 4), "write_factor": round(f_wr, 4),
              "per_dispatch_bytes": [int(f_rd * 1024 * a + f_wr * 1024 * b) for a, b in zip(cs["FETCH_SIZE"], cs["WRITE_SIZE"])][:16],
              "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/prof_pmc.sh %s), per logical launch of "
                        "the 1000-frame bench batch; factors calibrated on the 1 GiB k_copy_simple of the same run "
                        "(MI355X_MICROARCH.md: FETCH_SIZE reads half on gfx950)" % tag}
print(json.dumps(out, indent=1))
