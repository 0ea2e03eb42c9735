#!/bin/bash
# One GPU-box session for an A/B of library builds: parity tests of the default build first, then interleaved timing.
# usage: bash tools/ab_round.sh <tag> <lib> [<lib> ...]   (libs as for tools/ab_time.py)
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
tag=$1; shift
mkdir -p gpurun_out/r02
timeout -k 10 420 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/r02/pytest_$tag.log 2>&1
rc=$?
tail -3 gpurun_out/r02/pytest_$tag.log
[ $rc -ne 0 ] && { grep -E "FAILED|Error|assert" gpurun_out/r02/pytest_$tag.log | head -20; exit $rc; }
timeout -k 10 400 python -u tools/ab_time.py "$@" > gpurun_out/r02/ab_$tag.txt 2>&1 && cat gpurun_out/r02/ab_$tag.txt
