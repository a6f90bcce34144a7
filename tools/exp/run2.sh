#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/run2
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/run2/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/run2/pytest.log
bash tools/exp/cold.sh 2>&1 | grep -vE "slot: events 0|^fpv-mi355x: stats" | cut -c1-230
timeout -k 10 600 bash tests/e2e_cli.sh 2>&1 | grep -E "^threads=(8|32|64) total" 
