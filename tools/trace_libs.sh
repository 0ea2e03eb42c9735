#!/bin/bash
# rocprofv3 kernel statistics of the driver-form bench for each library build (no tests): bash tools/trace_libs.sh <tag> "<bench args>" <lib> ...
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
tag=$1; args=$2; shift 2
O=gpurun_out/r03/trace_$tag; mkdir -p $O
for l in "$@"; do
  if [ "$l" = default ]; then unset VSL_LIBRARY; else export VSL_LIBRARY=$PWD/tools/$l; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$l -o kt -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-extras $args > $O/$l.log 2>&1 || { tail -5 $O/$l.log; exit 1; }
  echo "== $l"; python tools/kernel_stats.py $O/$l
  find $O/$l -name '*kernel_trace.csv' -delete
done
