"""The header-only libtorch binding (include/vsloss_torch.h), compiled as tests/cpp/vsl_torch_ext.so,
against the reference fixtures: same tolerances as the ctypes path (loss 1e-5, gradients 1e-5 relative)."""
import importlib.util
import os

import numpy as np
import pytest
import torch

from golden_util import fid_key, load_case

EXT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpp", "vsl_torch_ext.so")


def load_ext():
    if not os.path.exists(EXT):
        pytest.fail("tests/cpp/vsl_torch_ext.so is not built (python -c 'import __graft_entry__ as g; g.build()')")
    spec = importlib.util.spec_from_file_location("vsl_torch_ext", EXT)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_extension_imports_without_gpu():
    assert hasattr(load_ext(), "loss")


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["b2_24x80_automask", "b2_24x80_stereo", "b1_20x50_single_source", "b2_24x80_scales4"])
def test_libtorch_binding_matches_fixture(name):
    ext = load_ext()
    cfg, z = load_case(name)
    dev = "cuda:0"
    srcs = [fid_key(f) for f in cfg["frame_ids"]][1:]
    disp = torch.from_numpy(z["disp"]).to(dev).requires_grad_(True)
    poses = [torch.from_numpy(z["T_%s" % f]).to(dev).requires_grad_(True) for f in srcs]
    noise = torch.from_numpy(z["noise"]).to(dev) if cfg["automask"] else torch.Tensor()
    losses, ident = ext.loss(disp, torch.from_numpy(z["image_0"]).to(dev),          # float64 images are cast on load
                             [torch.from_numpy(z["image_%s" % f]).to(dev) for f in srcs],
                             torch.from_numpy(z["K"]).to(dev), torch.from_numpy(z["inv_K"]).to(dev), poses, noise,
                             cfg["automask"], cfg["avg_reprojection"], cfg["no_ssim"], 1.0,
                             [1e-3 / 2 ** s for s in cfg["scales"]])
    losses[0].backward()
    ref = float(z["loss"])
    assert abs(losses[0].item() - ref) <= 1e-5 * ref
    rel = lambda a, b: float(np.linalg.norm(a.detach().cpu().numpy().astype(np.float64) - b) / np.linalg.norm(b))
    assert rel(disp.grad, z["grad_disp"]) <= 1e-5
    for i, f in enumerate(srcs):
        assert rel(poses[i].grad, z["grad_T_%s" % f]) <= 1e-5
    if cfg["automask"]:
        assert np.array_equal(ident.cpu().numpy(), z["identity_selection"])
