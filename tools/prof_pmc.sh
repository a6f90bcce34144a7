#!/bin/bash
# PMC passes for the bench workload (each pass its own run: SQ 8 slots, TCC 4;
# FETCH_SIZE and WRITE_SIZE do not fit one pass).  Results: gpurun_out/pmc_<tag>/
set -u
cd "$(dirname "$0")/.."
TAG=${1:-r01}
export TMPDIR=/tmp
run() {  # name counters...
  local name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc_$TAG/$name -- python3 bench.py --steps 2 --warmup 1 --quick > gpurun_out/pmc_$TAG/$name.log 2>&1
}
mkdir -p gpurun_out/pmc_$TAG
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
run sq2 SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
run fetch FETCH_SIZE
run write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
run tcc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
run tcc2 TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum
run tcc3 TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_TAG_STALL_sum
run tcc4 TCC_BUSY_sum TCC_CYCLE_sum TCC_STREAMING_REQ_sum TCC_NC_REQ_sum
run ta TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum
run grbm GRBM_GUI_ACTIVE GRBM_COUNT GRBM_UTCL2_BUSY GRBM_EA_BUSY GRBM_TC_BUSY GRBM_TA_BUSY
find gpurun_out/pmc_$TAG -name "*counter_collection.csv" | wc -l
