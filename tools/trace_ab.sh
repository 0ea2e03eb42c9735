#!/bin/bash
# GPU-box session: parity tests of the default build, then rocprofv3 kernel statistics of the driver-form bench for each library.
# usage: bash tools/trace_ab.sh <tag> <lib> [<lib> ...]   ("default" or a file under tools/)
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
tag=$1; shift
O=gpurun_out/r02/trace_$tag; mkdir -p $O
timeout -k 10 420 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $O/pytest.log 2>&1
rc=$?; tail -2 $O/pytest.log
[ $rc -ne 0 ] && { grep -E "FAILED|Error|assert" $O/pytest.log | head -20; exit $rc; }
for l in "$@"; do
  if [ "$l" = default ]; then unset VSL_LIBRARY; else export VSL_LIBRARY=$PWD/tools/$l; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$l -o kt -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/$l.log 2>&1 || { tail -5 $O/$l.log; exit 1; }
  echo "== $l"; python tools/kernel_stats.py $O/$l
  grep '^{' $O/$l.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step_us %.1f frames/s %.0f'%(d['ms_per_step']*1e3, d['value']))"
  find $O/$l -name '*kernel_trace.csv' -delete
done
