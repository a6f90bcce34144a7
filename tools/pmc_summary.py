#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc CSVs (tools/prof_pmc.sh) per kernel: mean counter
value per dispatch.  usage: pmc_summary.py gpurun_out/pmc_<tag> [kernel-substring]"""
import csv, glob, os, re, sys, collections, json
root = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"\b(k_[a-z0-9_]+)", r["Kernel_Name"])   # "void fpvhip::(anonymous namespace)::k_tile<false, true>(...)"
        k = m.group(1) if m else r["Kernel_Name"][:40]
        if flt in r["Kernel_Name"]:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in acc.items():
    out[k] = {c: sum(v) / len(v) for c, v in sorted(cs.items())}
    out[k]["_dispatches"] = max(len(v) for v in cs.values())
print(json.dumps(out, indent=1))
