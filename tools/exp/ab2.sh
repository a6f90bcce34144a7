#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/ab2
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/ab2/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/ab2/pytest.log
tail -3 gpurun_out/ab2/pytest.log
for kind in plasma noise mixed; do
  FPV_HIP_LIB=$PWD/build/ab/libfpvhip_r01.so timeout -k 10 300 python tools/ab_forward.py $kind 2>&1 | grep "^{" | tee -a gpurun_out/ab2/ab.log
  timeout -k 10 300 python tools/ab_forward.py $kind 2>&1 | grep "^{" | tee -a gpurun_out/ab2/ab.log
done
AB_NODELTA=1 FPV_HIP_LIB=$PWD/build/ab/libfpvhip_r01.so timeout -k 10 300 python tools/ab_forward.py plasma 2>&1 | grep "^{" | tee -a gpurun_out/ab2/ab.log
AB_NODELTA=1 timeout -k 10 300 python tools/ab_forward.py plasma 2>&1 | grep "^{" | tee -a gpurun_out/ab2/ab.log
for w in 4 6 10 12 16; do FPV_WAVES_PER_CU=$w timeout -k 10 300 python tools/ab_forward.py plasma 2>&1 | grep "^{" | sed "s/^/waves=$w /" | tee -a gpurun_out/ab2/ab.log; done
timeout -k 10 120 build/exp/pin_bench > gpurun_out/ab2/pin_bench.log 2>&1; tail -50 gpurun_out/ab2/pin_bench.log
