#!/bin/bash
# Hardware-counter passes over a short bench run (one pass per rocprofv3 run; chain stops at the first failure).
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/pmc_$1
mkdir -p $OUT
export TMPDIR=/tmp
B="bench.py --steps 3 --warmup 1 --no-cpu-baseline"
P1="TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE GRBM_TA_BUSY"
P2="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS"
P3="SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_BANK_CONFLICT TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum"
timeout -s KILL 120 rocprofv3 --pmc $P1 --output-format csv -d $OUT/p1 -o p -- python $B > $OUT/p1.log 2>&1 \
 && timeout -s KILL 120 rocprofv3 --pmc $P2 --output-format csv -d $OUT/p2 -o p -- python $B > $OUT/p2.log 2>&1 \
 && timeout -s KILL 120 rocprofv3 --pmc $P3 --output-format csv -d $OUT/p3 -o p -- python $B > $OUT/p3.log 2>&1 \
 && python tools/pmc_summary.py $OUT/p1 $OUT/p2 $OUT/p3 > $OUT/summary.txt ; echo "exit $?"; grep -A40 "^vsl_fused_kernel" $OUT/summary.txt | head -60
