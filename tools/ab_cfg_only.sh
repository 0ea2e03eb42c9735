#!/bin/bash
# interleaved timing of library builds on another benchmark configuration: bash tools/ab_cfg_only.sh <tag> "<bench args>" <lib> ...
cd "${GRAFT_REPO_ROOT:-.}"
tag=$1; shift
mkdir -p gpurun_out/r02
timeout -k 10 500 python -u tools/ab_cfg.py "$@" > gpurun_out/r02/ab_$tag.txt 2>&1; cat gpurun_out/r02/ab_$tag.txt
