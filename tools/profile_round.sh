#!/bin/bash
# Round profile: rocprofv3 kernel trace + stats of the default bench, then the
# PMC passes (separate runs, no tracing domains mixed in).  Output: gpurun_out/<tag>/
set -u
cd "$(dirname "$0")/.."
TAG=${1:-r01}
export TMPDIR=/tmp
mkdir -p gpurun_out/$TAG
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/trace -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/$TAG/bench_under_rocprof.log 2>&1
cp $(find gpurun_out/$TAG/trace -name "*kernel_stats.csv" | head -1) gpurun_out/$TAG/kernel_stats.csv
bash tools/prof_pmc.sh $TAG > /dev/null 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc_$TAG k_stream > gpurun_out/$TAG/pmc_k_stream.json
timeout 300 python3 bench.py > gpurun_out/$TAG/bench.log 2>&1
grep "^{" gpurun_out/$TAG/bench.log > gpurun_out/$TAG/bench.json
rm -rf gpurun_out/$TAG/trace
cat gpurun_out/$TAG/kernel_stats.csv | cut -c1-160 | head -8
