#!/usr/bin/env python3
"""HBM traffic per kernel launch from rocprofv3 --pmc passes (tools/prof_pmc.sh).

usage: traffic_from_pmc.py gpurun_out/pmc_<tag> > profiles/traffic.json

FETCH_SIZE / WRITE_SIZE are reported in KiB per dispatch.  Corrections, as
MI355X_MICROARCH.md (HBM / rocprofv3) prescribes for gfx950: FETCH_SIZE counts a
wide coalesced streaming read at half its bytes -> doubled; WRITE_SIZE is exact
for 16-B-per-lane streaming stores.  One k_stream dispatch per batch (round 2 on; round 1
launched a pair per batch)."""
import collections
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
            key = ("k_stream" if "k_stream" in name else "k_redo_stream" if "k_redo_stream" in name else
                   "k_inverse_spec" if "k_inverse_spec" in name else "k_inverse_trio" if "k_inverse_trio" in name else None)
            if key:
                per[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in per.items():
    group = 1                                     # dispatches per logical launch
    rd = 2.0 * 1024 * sum(cs["FETCH_SIZE"]) / (len(cs["FETCH_SIZE"]) / group)
    wr = 1024 * sum(cs["WRITE_SIZE"]) / (len(cs["WRITE_SIZE"]) / group)
    out[k] = {"bytes_per_launch": int(rd + wr), "read_bytes": int(rd), "write_bytes": int(wr),
              "dispatches_sampled": len(cs["FETCH_SIZE"]),
              "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/prof_pmc.sh %s), per logical launch of "
                        "the 1000-frame bench batch; FETCH_SIZE doubled per MI355X_MICROARCH.md gfx950 note" % tag}
print(json.dumps(out, indent=1))
