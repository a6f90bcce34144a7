#!/bin/bash
# GPU call: parity tests of the new forward path, then old-vs-new timing on the same box
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/ab1
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/ab1/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/ab1/pytest.log
tail -5 gpurun_out/ab1/pytest.log
for kind in plasma noise mixed; do
  FPV_HIP_LIB=$PWD/build/ab/libfpvhip_r01.so timeout -k 10 300 python tools/ab_forward.py $kind 2>&1 | grep "^{" | tee -a gpurun_out/ab1/ab.log
  timeout -k 10 300 python tools/ab_forward.py $kind 2>&1 | grep "^{" | tee -a gpurun_out/ab1/ab.log
done
