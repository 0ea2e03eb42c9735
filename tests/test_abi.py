"""CPU-side checks of the boundary: the shared library loads, exports every symbol the
header declares, the ctypes mirror matches the C struct, and argument validation works
without touching a GPU."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import mde_amd._abi as abi
    header = open(os.path.join(ROOT, "include", "vsloss.h")).read()
    declared = set(re.findall(r"\b(vsl_[a-z_]+)\s*\(", header))
    assert declared == set(abi.EXPORTS)
    for name in declared:
        assert hasattr(abi.lib, name), name
    assert abi.lib.vsl_version() >= 100


def test_descriptor_mirror_and_workspace_query():
    import mde_amd._abi as abi
    d = abi.LossDesc()
    assert ctypes.sizeof(d) == 208
    assert ctypes.sizeof(abi.Head) == 128
    d.B, d.H, d.W, d.S = 12, 192, 640, 2
    n = abi.lib.vsl_loss_workspace_bytes(ctypes.byref(d))
    assert n >= 2 * 12 * 192 * 640 * 4 and n % 256 == 0
    d.B = 0
    assert abi.lib.vsl_loss_workspace_bytes(ctypes.byref(d)) == 0


def test_argument_validation_without_gpu():
    import mde_amd._abi as abi
    d = abi.LossDesc()
    d.B, d.H, d.W, d.S = 1, 1, 8, 1          # H < 2
    rc = abi.lib.vsl_loss_forward(ctypes.byref(d), None, 0, None, None, None)
    assert rc == -1 and b"H >= 2" in abi.lib.vsl_last_error_string()
    d.H, d.S = 8, 9                            # too many sources
    assert abi.lib.vsl_loss_forward(ctypes.byref(d), None, 0, None, None, None) == -1
    d.S, d.n_scales, d.min_depth, d.max_depth = 1, 1, 0.1, 100.0
    assert abi.lib.vsl_loss_forward(ctypes.byref(d), None, 0, None, None, None) == -1   # NULL tensors
    with pytest.raises(abi.VslError):
        abi.check(-1)
