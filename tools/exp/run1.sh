#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/run1
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/run1/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/run1/pytest.log
timeout -k 10 600 python bench.py > gpurun_out/run1/bench.log 2>&1; echo "bench rc=$?"; grep "^{" gpurun_out/run1/bench.log > gpurun_out/run1/bench.json; tail -c 3000 gpurun_out/run1/bench.log
timeout -k 10 300 python tools/pcie_rate.py > gpurun_out/run1/pcie.log 2>&1; cat gpurun_out/run1/pcie.log | tail -5
timeout -k 10 600 bash tests/e2e_cli.sh > gpurun_out/run1/e2e_cli.log 2>&1; grep -E "threads=|ok|rror" gpurun_out/run1/e2e_cli.log | head -40
