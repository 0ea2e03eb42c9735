#!/bin/bash
# Instruction-fetch / instruction-cache and instruction-class counters of a short bench run (one pass per rocprofv3 run).
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/pmc_ifetch
mkdir -p $OUT
export TMPDIR=/tmp
B="bench.py --steps 3 --warmup 1 --no-cpu-baseline"
P1="SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_ICACHE_BUSY_CYCLES SQC_TC_INST_REQ SQ_BUSY_CYCLES"
P2="SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"
timeout -s KILL 200 rocprofv3 --pmc $P1 --output-format csv -d $OUT/p1 -o p -- python $B > $OUT/p1.log 2>&1 \
 && timeout -s KILL 200 rocprofv3 --pmc $P2 --output-format csv -d $OUT/p2 -o p -- python $B > $OUT/p2.log 2>&1 \
 && python tools/pmc_summary.py $OUT/p1 $OUT/p2 > $OUT/summary.txt ; echo "exit $?"; grep -A9 "^vsl_fused_kernel" $OUT/summary.txt
