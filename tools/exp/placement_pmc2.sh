#!/bin/bash
# Memory-side counters of k_stream on several buffer sets (see placement_pmc.py): request
# latency (LEVEL / REQ), credit stalls, and the kernel's own duration under the profiler.
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
N=${1:-8}
mkdir -p gpurun_out/plpmc2
i=0
for grp in "GRBM_GUI_ACTIVE TCC_BUSY_sum TCC_TAG_STALL_sum" \
           "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum" \
           "TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" \
           "TCC_EA0_RDREQ_GMI_CREDIT_STALL_sum TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d gpurun_out/plpmc2/g$i -- python3 tools/exp/placement_pmc.py $N > gpurun_out/plpmc2/run$i.log 2>&1
  grep "^{" gpurun_out/plpmc2/run$i.log | tail -1
  f=$(find gpurun_out/plpmc2/g$i -name "*counter_collection.csv" | head -1)
  python3 - "$f" $N <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_stream" in r["Kernel_Name"]]
by = collections.OrderedDict()
for r in rows:
    by.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
for did, c in list(by.items())[-int(sys.argv[2]):]:
    print(did, {k: int(v) for k, v in c.items()})
PY
done
