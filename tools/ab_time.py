"""A/B timing of library builds (GPU box): interleaved rounds in one process per library is not
possible with ctypes name clashes, so each library runs in a subprocess; several rounds each."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sys.argv[1:]
res = {l: [] for l in libs}
for rnd in range(3):
    for l in libs:
        env = dict(os.environ)
        if l != "default":
            env["VSL_LIBRARY"] = os.path.join(ROOT, "tools", l)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "40", "--warmup", "10", "--no-cpu-baseline", "--no-e2e", "--no-extras"],
                             env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
        d = json.loads(out)
        res[l].append((d["roofline"]["kernel_ms"] * 1e3, d["ms_per_step"] * 1e3))
for l in libs:
    print("%-22s fused_us %s   step_us %s" % (l, " ".join("%.1f" % a for a, _ in res[l]), " ".join("%.0f" % b for _, b in res[l])))
