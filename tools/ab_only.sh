#!/bin/bash
# interleaved timing of library builds only (no tests): bash tools/ab_only.sh <tag> <lib> [<lib> ...]
cd "${GRAFT_REPO_ROOT:-.}"
tag=$1; shift
mkdir -p gpurun_out/r03
timeout -k 10 500 python -u tools/ab_time.py "$@" > gpurun_out/r03/ab_$tag.txt 2>&1; cat gpurun_out/r03/ab_$tag.txt
