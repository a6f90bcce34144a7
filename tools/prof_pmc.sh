#!/bin/bash
# PMC passes for the bench workload (each pass its own run: SQ 8 slots, TCC 4;
# FETCH_SIZE and WRITE_SIZE do not fit one pass; --pmc only, never with a trace domain).
# The profiled program is bench.py's --pmc-child: three forward batches, two inverse batches
# and a 1 GiB calibration copy.  Results: gpurun_out/pmc_<tag>/ (+ _noise: the adversarial workload)
set -u
cd "$(dirname "$0")/.."
TAG=${1:-r05}
export TMPDIR=/tmp
run() {  # dir name counters...
  local dir=$1 name=$2; shift 2
  timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $dir/$name -- python3 bench.py --pmc-child $EXTRA > $dir/$name.log 2>&1
}
mkdir -p gpurun_out/pmc_$TAG gpurun_out/pmc_${TAG}_noise
EXTRA=""
D=gpurun_out/pmc_$TAG
run $D sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
run $D sq2 SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
run $D fetch FETCH_SIZE
run $D write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
run $D tcc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
run $D ta TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum
run $D grbm GRBM_GUI_ACTIVE GRBM_COUNT
EXTRA="--workload noise"
D=gpurun_out/pmc_${TAG}_noise
run $D fetch FETCH_SIZE
run $D write WRITE_SIZE
find gpurun_out/pmc_$TAG gpurun_out/pmc_${TAG}_noise -name "*counter_collection.csv" | wc -l
