#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r03diag; mkdir -p $O
VSL_BENCH_WATCHDOG=40 VSL_BENCH_DEVICE=0 VSL_BENCH_BACKEND=gloo timeout -k 10 160 python -u bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --e2e-steps 3 > $O/n2.json 2> $O/n2.err
echo rc=$?; tail -60 $O/n2.err; cut -c1-600 $O/n2.json
