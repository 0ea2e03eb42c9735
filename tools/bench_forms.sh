#!/bin/bash
# GPU-box session: the bench in the driver's 20-step form and in the default 200-step form, three rounds interleaved.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r02
: > gpurun_out/r02/bench_forms.txt
for r in 1 2 3; do
  for f in "--steps 20 --warmup 5" "--steps 200 --warmup 20"; do
    timeout -k 10 120 python -u bench.py $f --no-cpu-baseline > gpurun_out/r02/bf_tmp.json 2> gpurun_out/r02/bf_tmp.err || { tail -5 gpurun_out/r02/bf_tmp.err; exit 1; }
    python - "$f" <<'PY' >> gpurun_out/r02/bench_forms.txt
import json, sys
d = json.loads(open("gpurun_out/r02/bf_tmp.json").read().strip().splitlines()[-1])
print("%-24s step_us %.1f  frames/s %.0f  fused_us %.1f  frac %.4f" % (sys.argv[1], d["ms_per_step"] * 1e3, d["value"], d["roofline"]["kernel_ms"] * 1e3, d["roofline"]["frac"]))
PY
  done
done
cat gpurun_out/r02/bench_forms.txt
