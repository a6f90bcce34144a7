#!/usr/bin/env python3
"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace CSV.

  python tools/trace_gaps.py <dir-or-kernel_trace.csv> [kernel-name-prefix]

Prints, per (previous kernel -> next kernel) pair, the number of hand-overs and the mean /
median / max gap (start of the next minus end of the previous, in microseconds), then the
per-kernel durations.  Gaps above 1 ms (the host was elsewhere) are counted separately."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

import numpy as np


def short(name):
    m = re.search(r"\b(k_[a-z0-9_]+)", name)
    return m.group(1) if m else name[:40]


def main():
    src = sys.argv[1]
    if os.path.isdir(src):
        src = glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = []
    for r in csv.DictReader(open(src)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    gaps, dur, long_gaps = defaultdict(list), defaultdict(list), defaultdict(int)
    for (s0, e0, k0), (s1, e1, k1) in zip(rows, rows[1:]):
        g = (s1 - e0) / 1e3
        if g > 1000.0:
            long_gaps[(k0, k1)] += 1
        else:
            gaps[(k0, k1)].append(g)
    for s, e, k in rows:
        dur[k].append((e - s) / 1e3)
    print(f"{'previous -> next':58s} {'n':>6s} {'mean':>8s} {'median':>8s} {'max':>8s}  (us)")
    for (k0, k1), v in sorted(gaps.items(), key=lambda kv: -len(kv[1])):
        a = np.array(v)
        print(f"{k0 + ' -> ' + k1:58s} {len(a):6d} {a.mean():8.2f} {np.median(a):8.2f} {a.max():8.2f}"
              + (f"  (+{long_gaps[(k0, k1)]} above 1 ms)" if long_gaps[(k0, k1)] else ""))
    print()
    print(f"{'kernel':34s} {'n':>6s} {'mean':>9s} {'median':>9s} {'min':>9s}  (us)")
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        a = np.array(v)
        print(f"{k:34s} {len(a):6d} {a.mean():9.2f} {np.median(a):9.2f} {a.min():9.2f}")


if __name__ == "__main__":
    main()
