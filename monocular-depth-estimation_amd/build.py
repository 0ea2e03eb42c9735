"""Builds libvsloss.so in-tree with hipcc for gfx950 (no GPU needed to compile)."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libvsloss.so")
SOURCES = ["vsl_loss.hip", "vsl_layers.hip", "vsl_pose.hip", "vsl_metrics.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "vsloss.h")]
    return any(os.path.getmtime(d) > t for d in deps if os.path.isfile(d))


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return OUT
    # -ffp-contract=on: fuse a*b+c only inside one expression.  The default (fast) also fuses
    # across statements, which measurably moves the pose/disparity gradients away from the
    # reference's float32 results (1.5e-5 instead of 1e-6 relative on the golden fixtures).
    # -fno-slp-vectorize: packed f32 math has no rate advantage on gfx950 and its operand pairing costs
    # register moves (measured 3 % slower with it).
    cmd = [HIPCC, "-O3", "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=on", "-fno-slp-vectorize", "-shared", "-fPIC",
           "-I", os.path.join(HERE, "..", "include"), "-o", OUT] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
