"""A/B timing of library builds on other benchmark configurations (GPU box): tools/ab_cfg.py "<bench args>" lib...
Every library runs bench.py in a subprocess, three interleaved rounds."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args, libs = sys.argv[1].split(), sys.argv[2:]
res = {l: [] for l in libs}
for rnd in range(3):
    for l in libs:
        env = dict(os.environ)
        if l != "default":
            env["VSL_LIBRARY"] = os.path.join(ROOT, "tools", l)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "40", "--warmup", "10", "--no-cpu-baseline"] + args,
                             env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
        d = json.loads(out)
        res[l].append((d["roofline"]["kernel_ms"] * 1e3, d["ms_per_step"] * 1e3))
for l in libs:
    print("%-22s fused_us %s   step_us %s" % (l, " ".join("%.1f" % a for a, _ in res[l]), " ".join("%.0f" % b for _, b in res[l])))
