#!/bin/bash
# rocprofv3 kernel statistics of one benchmark configuration: bash tools/trace_cfg.sh <config>
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r02/trace_$1; mkdir -p $O
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o kt -- python bench.py --config $1 --steps 20 --warmup 5 --no-cpu-baseline > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
python tools/kernel_stats.py $O
find $O -name '*kernel_trace.csv' -delete
