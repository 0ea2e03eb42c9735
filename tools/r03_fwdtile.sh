#!/bin/bash
# GPU-box session: GPU tests, then the forward-only program's time (4 heads 640x192 b12; c5 kernel statistics).
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r03fwd; mkdir -p $O
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > $O/pytest_gpu.log 2>&1 || { tail -40 $O/pytest_gpu.log; exit 1; }
tail -2 $O/pytest_gpu.log
timeout -k 10 120 python -u tools/time_nograd.py 2>&1 | tail -2 || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5 -o kt -- python bench.py --config c5 --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-extras > $O/c5.log 2>&1 || { tail -5 $O/c5.log; exit 1; }
python tools/kernel_stats.py $O/c5; find $O -name '*kernel_trace.csv' -delete
tail -1 $O/c5.log | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('c5 step ms', r['ms_per_step'], 'frac', r['roofline']['frac'])"
