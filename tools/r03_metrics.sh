#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r03m; mkdir -p $O
timeout -k 10 300 python -u -m pytest tests/test_gpu_depth_metrics.py tests/test_gpu_train_step.py -m gpu -x -q --timeout 120 --timeout-method thread > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
timeout -k 10 120 python -u tools/time_metrics.py > $O/time_metrics.txt 2>&1 || { tail $O/time_metrics.txt; exit 1; }
cat $O/time_metrics.txt
