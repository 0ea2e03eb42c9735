#!/bin/bash
# GPU-box session: the driver's N>1 form (torch.distributed.run) rehearsed with two ranks on one card over gloo.
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r03tr; mkdir -p $O
VSL_BENCH_DEVICE=0 VSL_BENCH_BACKEND=gloo timeout -k 10 280 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 \
  bench.py --gpus 2 --steps 20 --warmup 5 > $O/n2.log 2>&1 || { tail -30 $O/n2.log; exit 1; }
grep '^{' $O/n2.log | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('n_gpus', r['n_gpus'], 'ranks_seen', r['ranks_seen'], 'value %.0f' % r['value'], 'rccl', r['rccl']['allreduce_ms'], 'e2e', r['e2e']['frames_per_s'], 'ddp', r['e2e']['ddp'])"
