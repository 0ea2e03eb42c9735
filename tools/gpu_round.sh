#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, kernel-trace stats and the two HBM counter passes.
# Every GPU step has its own time limit and the chain stops at the first failure.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r02prof
mkdir -p $OUT
export TMPDIR=/tmp
B="--steps 20 --warmup 5 --no-cpu-baseline"
timeout -k 10 420 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $OUT/pytest_gpu.log 2>&1 \
 && tail -3 $OUT/pytest_gpu.log \
 && timeout -k 10 120 python -u -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1 \
 && tail -1 $OUT/smoke.log \
 && timeout -k 10 240 python -u bench.py > $OUT/bench.json 2> $OUT/bench.err && timeout -k 10 120 python -u bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver_form.json 2>> $OUT/bench.err \
 && cat $OUT/bench.json \
 && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ktrace -o kt -- python bench.py $B > $OUT/ktrace.log 2>&1 \
 && timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pf -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1 \
 && timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pw -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_write.log 2>&1 \
 && python tools/pmc_summary.py --traffic 640x192_b12_S2_heads4 $OUT/pmc_traffic.json $OUT/pmc_fetch $OUT/pmc_write > $OUT/pmc_traffic.log 2>&1 \
 && find $OUT/ktrace -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \; \
 && python tools/trace_percentiles.py $OUT/ktrace vsl_fused_kernel > $OUT/fused_trace_percentiles.txt \
 && find $OUT/ktrace -name '*kernel_trace.csv' -delete \
 && find $OUT/pmc_fetch $OUT/pmc_write -name '*counter_collection.csv' -size +8M -delete ; \
 echo "exit chain: $?"; ls -la $OUT
