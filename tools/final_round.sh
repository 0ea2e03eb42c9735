#!/bin/bash
# Final GPU-box session of a round: tools/gpu_round.sh (tests, smoke, bench in both forms, kernel trace, HBM counters), then the other
# benchmark shapes and the pipe counters.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
bash tools/gpu_round.sh || exit 1
grep -q "exit chain: 0" /dev/null
OUT=gpurun_out/r02prof
: > $OUT/other_configs.jsonl
for c in c1 c3 c5; do
  timeout -k 10 150 python -u bench.py --config $c --steps 100 --warmup 10 --no-cpu-baseline >> $OUT/other_configs.jsonl 2>> $OUT/bench.err || exit 1
done
python - <<'PY'
import json
for l in open("gpurun_out/r02prof/other_configs.jsonl"):
    d = json.loads(l)
    print("%-70s step %.1f us  fused %.1f us  frac %.3f  %.0f frames/s" % (d["metric"][-46:], d["ms_per_step"] * 1e3, d["roofline"]["kernel_ms"] * 1e3, d["roofline"]["frac"], d["value"]))
PY
bash tools/pmc_passes.sh final > $OUT/pmc_pipes.log 2>&1; tail -45 $OUT/pmc_pipes.log
