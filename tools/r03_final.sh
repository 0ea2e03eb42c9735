#!/bin/bash
# GPU-box session: what the driver runs at round end (GPU tests, smoke, default bench) plus the driver-form bench and
# the two-rank rehearsal.  Every GPU step has its own time limit; the chain stops at the first failure.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r03final; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > $O/pytest_gpu.log 2>&1 \
  || { tail -30 $O/pytest_gpu.log; exit 1; }
tail -2 $O/pytest_gpu.log
timeout -k 10 120 python -u -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail $O/smoke.log; exit 1; }
tail -2 $O/smoke.log
T0=$(date +%s.%N)
timeout -k 10 300 python -u bench.py > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python -c "import sys; print('default bench wall seconds: %.1f' % (float(sys.argv[2]) - float(sys.argv[1])))" $T0 $(date +%s.%N)
timeout -k 10 200 python -u bench.py --steps 20 --warmup 5 > $O/bench_driver_form.json 2>> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python - <<'PY'
import json
for f in ("bench.json", "bench_driver_form.json"):
    r = json.loads(open("gpurun_out/r03final/" + f).read().strip().splitlines()[-1])
    print(f, "value %.0f ms %.4f frac %.4f kernel_us %.1f" % (r["value"], r["ms_per_step"], r["roofline"]["frac"], r["roofline"]["kernel_ms"] * 1e3),
          "e2e %.0f f/s loss_ms %.3f" % (r["e2e"]["frames_per_s"], r["e2e"]["loss_ms_per_step"]),
          "c3 %.3f c5 %.3f smooth %.3f" % (r["other_configs"][0]["frac"], r["other_configs"][1]["frac"], r["smooth_disp"]["frac"]),
          "cpu %.2f f/s x%.0f in %.1fs" % (r["cpu_baseline"]["value"], r["vs_cpu_baseline"], r["cpu_baseline"]["seconds"]))
PY
VSL_BENCH_DEVICE=0 VSL_BENCH_BACKEND=gloo timeout -k 10 300 python -u bench.py --gpus 2 --steps 20 --warmup 5 --e2e-steps 5 > $O/bench_n2.json 2> $O/bench_n2.err \
  || { tail -20 $O/bench_n2.err; exit 1; }
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r03final/bench_n2.json").read().strip().splitlines()[-1])
print("n2: n_gpus", r["n_gpus"], "ranks_seen", r["ranks_seen"], "rccl", r["rccl"], "e2e f/s", r["e2e"]["frames_per_s"])
PY
