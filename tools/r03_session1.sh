#!/bin/bash
# GPU-box session: GPU parity tests, the default bench line, bench.py --gpus 2 started without a launcher (both ranks
# on card 0, gloo) with the end-to-end block.  Every GPU step has its own time limit; the chain stops at the first failure.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r03s1; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > $O/pytest_gpu.log 2>&1 \
  || { tail -30 $O/pytest_gpu.log; exit 1; }
tail -3 $O/pytest_gpu.log
timeout -k 10 300 python -u bench.py > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r03s1/bench.json").read().strip().splitlines()[-1])
print("value", r["value"], "ms", r["ms_per_step"], "frac", r["roofline"]["frac"], "kernel_ms", r["roofline"]["kernel_ms"])
print("e2e", r.get("e2e"))
print("other", r.get("other_configs"))
print("smooth", r.get("smooth_disp"))
cb = r.get("cpu_baseline", {})
print("cpu", {k: cb.get(k) for k in ("value", "cores", "cpu_model", "os_cpu_count", "seconds", "thread_trials_ms", "gpu_over_cpu")})
PY
VSL_BENCH_DEVICE=0 VSL_BENCH_BACKEND=gloo timeout -k 10 300 python -u bench.py --gpus 2 --steps 20 --warmup 5 --e2e-steps 5 > $O/bench_n2.json 2> $O/bench_n2.err \
  || { tail -20 $O/bench_n2.err; exit 1; }
cut -c1-1500 $O/bench_n2.json
