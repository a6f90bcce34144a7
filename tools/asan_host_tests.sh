#!/bin/bash
# Host codec (container format, batching, worker pool, decoders, Frame API glue)
# under AddressSanitizer + UBSan, over the CPU test shim.  CPU only.
set -e
cd "$(dirname "$0")/.."
make asan
export FPV_TEST_SHIM=$PWD/build/asan/libfpv_amd_cpushim.so
export LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
exec python -m pytest tests/test_host_cpu.py tests/test_frame_api.py -m "not gpu" -x -q "$@"
