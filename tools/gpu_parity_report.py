"""Print the parity figures of the HIP path for every golden fixture (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from golden_util import case_names
from test_gpu_fused_loss import run_fixture, rel

for name in case_names():
    cfg, z, srcs, losses, ident, gdisp, gposes = run_fixture(name)
    s = "%-34s loss=%.2e smooth=%.2e gdisp=%.2e" % (name, abs(losses[0] - float(z["loss"])) / float(z["loss"]),
                                                    abs(losses[2] - float(z["smooth"])) / float(z["smooth"]), rel(gdisp, z["grad_disp"]))
    for i, f in enumerate(srcs):
        s += " gT_%s=%.2e" % (f, rel(gposes[i], z["grad_T_%s" % f]))
    if cfg["automask"]:
        s += " flips=%d" % int((ident.cpu().numpy() != z["identity_selection"]).sum())
    print(s)
