#!/bin/bash
# HBM traffic of the tile pass at 4096x4096 x 60 (FETCH_SIZE and WRITE_SIZE in separate --pmc passes, calibrated on the copy):
# bash tools/exp/large_pmc.sh [xs ys n]   -> gpurun_out/pmc_large/traffic.json
cd "$(dirname "$0")/../.." && export TMPDIR=/tmp
D=gpurun_out/pmc_large; rm -rf $D; mkdir -p $D
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/fetch -- python3 tools/exp/large_pmc.py "$@" > $D/fetch.log 2>&1 &&
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/write -- python3 tools/exp/large_pmc.py "$@" > $D/write.log 2>&1 &&
python3 tools/traffic_from_pmc.py $D > $D/traffic.json
rm -rf $D/*/*/*.db 2>/dev/null
cat $D/traffic.json
