#!/bin/bash
# GPU-box session: kernel-trace statistics of the driver-form bench, the two HBM traffic passes, and the
# vector-memory counters for the blurred-noise and the piecewise-smooth disparity.  One rocprofv3 run per pass.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r03prof
mkdir -p $OUT
export TMPDIR=/tmp
Q="--no-cpu-baseline --no-e2e --no-extras"
B="bench.py --steps 20 --warmup 5 $Q"
S="bench.py --steps 3 --warmup 1 $Q"
P1="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_TA_BUSY_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE"
P3="SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ktrace -o kt -- python $B > $OUT/ktrace.log 2>&1 \
 && timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pf -- python $S > $OUT/pmc_fetch.log 2>&1 \
 && timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pw -- python $S > $OUT/pmc_write.log 2>&1 \
 && python tools/pmc_summary.py --traffic 640x192_b12_S2_heads4 $OUT/pmc_traffic.json $OUT/pmc_fetch $OUT/pmc_write > $OUT/pmc_traffic.log 2>&1 \
 && timeout -s KILL 200 rocprofv3 --pmc $P1 --output-format csv -d $OUT/p1_noise -o p -- python $S > $OUT/p1_noise.log 2>&1 \
 && timeout -s KILL 200 rocprofv3 --pmc $P1 --output-format csv -d $OUT/p1_smooth -o p -- python $S --disp smooth > $OUT/p1_smooth.log 2>&1 \
 && timeout -s KILL 200 rocprofv3 --pmc $P3 --output-format csv -d $OUT/p3 -o p -- python $S > $OUT/p3.log 2>&1 \
 && python tools/pmc_summary.py $OUT/p1_noise > $OUT/p1_noise.txt && python tools/pmc_summary.py $OUT/p1_smooth > $OUT/p1_smooth.txt \
 && python tools/pmc_summary.py $OUT/p3 > $OUT/p3.txt \
 && find $OUT/ktrace -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \; \
 && python tools/trace_percentiles.py $OUT/ktrace vsl_fused_kernel > $OUT/fused_trace_percentiles.txt \
 && find $OUT -name '*kernel_trace.csv' -delete \
 && find $OUT -name '*counter_collection.csv' -size +4M -delete ; \
 echo "exit chain: $?"; cat $OUT/kernel_stats.csv | cut -c1-200 | head -8; grep -A8 "^vsl_fused_kernel" $OUT/p1_noise.txt $OUT/p1_smooth.txt $OUT/p3.txt; cat $OUT/pmc_traffic.log | head -30
