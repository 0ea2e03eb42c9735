"""`python bench.py --gpus 2` with no launcher around it starts its own two rank processes and reports what the
process group saw.  On a one-GPU box both ranks share card 0 and gloo stands in for RCCL (which refuses two ranks
on one device): the rehearsal form of the 8-GPU run of BASELINE configuration 4."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env.update(extra)
    return env


@pytest.mark.gpu
def test_bench_starts_its_own_ranks():
    proc = subprocess.run(
        [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
         "--no-cpu-baseline", "--no-e2e"],
        env=_clean_env(VSL_BENCH_DEVICE="0", VSL_BENCH_BACKEND="gloo"), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
        text=True, timeout=300)
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, proc.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["ranks_seen"] == 2
    assert rec["config"]["global_batch"] == 24 and rec["scaling"] == "weak"
    assert rec["rccl"]["bytes"] == 4 * 13_400_000 and rec["rccl"]["allreduce_ms"] > 0
    assert rec["value"] > 0 and rec["roofline"]["frac"] > 0 and "cpu_baseline" not in rec


def test_bench_refuses_more_ranks_than_gpus():
    """Without enough GPUs (none in a CPU container) the parent says so and starts nothing."""
    import torch
    if torch.cuda.device_count() >= 64:
        pytest.skip("a box with 64 GPUs")
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "1"],
                          env=_clean_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert proc.returncode == 2
    assert "GPU(s) visible" in proc.stderr and proc.stdout.strip() == ""


def test_bench_reports_a_failing_rank():
    """Ranks that cannot come up (no GPU behind the device they are pinned to) end the run with a non-zero exit code
    and a message, quickly -- the parent does not sit waiting for a line that will never come."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"],
                          env=_clean_env(VSL_BENCH_DEVICE="0", VSL_BENCH_BACKEND="gloo"), stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, text=True, timeout=180)
    assert proc.returncode != 0
    assert "a rank failed" in proc.stderr and not [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
