#!/bin/bash
# GPU-box session: the GPU parity tests only.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r03t; mkdir -p $O
timeout -k 10 700 python -u -m pytest ${PYTEST_FILES:-tests} -m gpu -x -q --timeout 180 --timeout-method thread ${PYTEST_ARGS} > $O/pytest_gpu.log 2>&1
rc=$?
tail -${TAILN:-12} $O/pytest_gpu.log
exit $rc
