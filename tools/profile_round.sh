#!/bin/bash
# Round profile: rocprofv3 kernel trace + stats of the bench (quick mode: the headline workload
# and the inverse, enough launches for the averages to mean something), then the PMC passes
# (separate runs, no tracing domain mixed in), then the plain default bench.  Output: gpurun_out/<tag>/
set -u
cd "$(dirname "$0")/.."
TAG=${1:-r05}
export TMPDIR=/tmp
mkdir -p gpurun_out/$TAG
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/trace -- python3 bench.py --steps 200 --warmup 5 --quick > gpurun_out/$TAG/bench_under_rocprof.log 2>&1
cp $(find gpurun_out/$TAG/trace -name "*kernel_stats.csv" | head -1) gpurun_out/$TAG/kernel_stats.csv
echo "[$(date +%T)] trace done"
bash tools/prof_pmc.sh $TAG > /dev/null 2>&1
echo "[$(date +%T)] pmc done"
python3 tools/pmc_summary.py gpurun_out/pmc_$TAG k_tile > gpurun_out/$TAG/pmc_k_tile.json
python3 tools/pmc_summary.py gpurun_out/pmc_$TAG k_inverse_ > gpurun_out/$TAG/pmc_k_inverse.json
python3 tools/traffic_from_pmc.py gpurun_out/pmc_$TAG > gpurun_out/$TAG/traffic.json
python3 tools/traffic_from_pmc.py gpurun_out/pmc_${TAG}_noise > gpurun_out/$TAG/traffic_noise.json
timeout -k 10 500 python3 bench.py > gpurun_out/$TAG/bench.log 2>&1
grep "^{" gpurun_out/$TAG/bench.log > gpurun_out/$TAG/bench.json
rm -rf gpurun_out/$TAG/trace gpurun_out/pmc_$TAG/*/*/*.db gpurun_out/pmc_${TAG}_noise/*/*/*.db 2>/dev/null
cat gpurun_out/$TAG/kernel_stats.csv | cut -c1-160 | head -8
cat gpurun_out/$TAG/traffic.json | head -40
