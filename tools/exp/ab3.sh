#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/ab3
timeout -k 10 600 python -m pytest tests/test_gpu_predict.py tests/test_gpu_golden.py tests/test_gpu_codec.py -m gpu -x -q > gpurun_out/ab3/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/ab3/pytest.log
tail -3 gpurun_out/ab3/pytest.log
for rep in 1 2; do
for lib in build/ab/libfpvhip_r01.so build/ab/libfpvhip_abl0.so fusion-power-video_amd/lib/libfpvhip.so; do
  FPV_HIP_LIB=$PWD/$lib timeout -k 10 300 python tools/ab_forward.py plasma 2>&1 | grep "^{" | tee -a gpurun_out/ab3/ab.log
done; done
AB_NODELTA=1 timeout -k 10 300 python tools/ab_forward.py plasma 2>&1 | grep "^{" | tee -a gpurun_out/ab3/ab.log
timeout -k 10 300 python tools/ab_forward.py noise 2>&1 | grep "^{" | tee -a gpurun_out/ab3/ab.log
