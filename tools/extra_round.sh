#!/bin/bash
# GPU-box session: two-rank rehearsal of bench.py on one card (gloo instead of RCCL, both ranks on GPU 0) and the end-to-end driver.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r02extra; mkdir -p $O
VSL_BENCH_DEVICE=0 VSL_BENCH_BACKEND=gloo timeout -k 10 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
  bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_n2_rehearsal.log 2>&1 || { tail -20 $O/bench_n2_rehearsal.log; exit 1; }
tail -1 $O/bench_n2_rehearsal.log | cut -c1-400
timeout -k 10 240 python -u examples/synthetic_train.py --steps 20 > $O/e2e.log 2>&1 || { tail -20 $O/e2e.log; exit 1; }
tail -1 $O/e2e.log
timeout -k 10 240 python -u examples/synthetic_train.py --steps 20 --amp > $O/e2e_amp.log 2>&1 || { tail -20 $O/e2e_amp.log; exit 1; }
tail -1 $O/e2e_amp.log
