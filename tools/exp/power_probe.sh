#!/bin/bash
# Socket power and clocks (rocm-smi, every ~0.3 s, with wall-clock stamps) beside tools/exp/power_probe.py; amd-smi's
# throttle / violation counters before, between and after.
# bash tools/exp/power_probe.sh   -> gpurun_out/power_probe.log, gpurun_out/power_samples.log, gpurun_out/power_throttle.log
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out
rocm-smi --showpower --showclocks --showmaxpower > gpurun_out/power_samples.log 2>&1
{ amd-smi metric --help; echo ==== ; amd-smi metric -g 0 --throttle; echo ====; amd-smi metric -g 0 --power --clock --temperature; } > gpurun_out/power_throttle.log 2>&1
( while [ ! -e gpurun_out/power_done ]; do
    echo "t $(date +%s.%N | cut -c1-13)" >> gpurun_out/power_samples.log
    rocm-smi --showpower --showclocks --csv 2>/dev/null | grep -v "^$" >> gpurun_out/power_samples.log
    sleep 0.2
  done ) &
sampler=$!
( while [ ! -e gpurun_out/power_done ]; do
    echo "t $(date +%s.%N | cut -c1-13)" >> gpurun_out/power_throttle.log
    amd-smi metric -g 0 --throttle --json >> gpurun_out/power_throttle.log 2>&1
    sleep 1
  done ) &
sampler2=$!
python tools/exp/power_probe.py > gpurun_out/power_probe.log 2>&1
rc=$?
touch gpurun_out/power_done
wait $sampler $sampler2
rm -f gpurun_out/power_done
exit $rc
