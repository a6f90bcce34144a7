#!/bin/bash
# TLB-side counters of k_stream on several buffer sets (see placement_pmc.py)
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out/plpmc
i=0
for grp in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" \
           "TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_LFIFO_FULL_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d gpurun_out/plpmc/g$i -- python3 tools/exp/placement_pmc.py 5 > gpurun_out/plpmc/run$i.log 2>&1
  tail -1 gpurun_out/plpmc/run$i.log
  f=$(find gpurun_out/plpmc/g$i -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_stream" in r["Kernel_Name"]]
by = collections.OrderedDict()
for r in rows:
    by.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
last = list(by.items())[-5:]
for did, c in last:
    print(did, {k: int(v) for k, v in c.items()})
PY
done
