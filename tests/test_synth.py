"""The synthetic inputs are pure functions of (seed, shape): the development container and the GPU box must build the
same tensors without sharing files.  Also the piecewise-smooth disparity of `bench.py --disp smooth`."""
import numpy as np
import torch


def test_make_batch_is_deterministic_and_in_range():
    import mde_amd.synth as synth
    a = synth.make_batch(2, 16, 40, [0, -1, 1, "s"], seed=7, n_heads=2)
    b = synth.make_batch(2, 16, 40, [0, -1, 1, "s"], seed=7, n_heads=2)
    for k in a:
        assert torch.equal(a[k], b[k]), k
    c = synth.make_batch(2, 16, 40, [0, -1, 1, "s"], seed=8, n_heads=2)
    assert not torch.equal(a[("disp", 0)], c[("disp", 0)])
    assert not torch.equal(a[("disp", 0)], a[("disp", 1)])
    for h in range(2):
        d = a[("disp", h)]
        assert d.dtype == torch.float32 and tuple(d.shape) == (2, 1, 16, 40) and 0.05 <= float(d.min()) and float(d.max()) <= 0.95
    assert float(a["stereo_T"][0, 0, 3]) == np.float32(0.1)
    assert tuple(a["noise"].shape) == (2, 3, 16, 40)
    K, iK = a["K"][0].double().numpy(), a["inv_K"][0].double().numpy()
    assert np.allclose(K @ iK, np.eye(4), atol=1e-6)          # inv_K = pinv(K), data_loaders.py:396


def test_smooth_disparity_is_piecewise_smooth_and_differs_between_heads():
    import mde_amd.synth as synth
    d0 = synth.smooth_disp(3, 96, 320, seed=1, head=0)
    d1 = synth.smooth_disp(3, 96, 320, seed=1, head=1)
    assert torch.equal(d0, synth.smooth_disp(3, 96, 320, seed=1, head=0))
    assert tuple(d0.shape) == (3, 1, 96, 320) and 0.05 <= float(d0.min()) and float(d0.max()) <= 0.95
    assert not torch.equal(d0, d1)
    # horizontally constant except at the few patch borders: nearly every horizontal neighbour pair is equal
    same = (d0[..., 1:] == d0[..., :-1]).float().mean().item()
    assert same > 0.97, same
    # the ground plane: below the horizon the disparity grows with the image row
    col = d0[0, 0, :, 0]
    assert float(col[-1]) > float(col[0]) and bool((col[1:] >= col[:-1] - 1e-6).float().mean() > 0.9)
    batch = synth.make_batch(1, 32, 64, [0, -1, 1], seed=3, n_heads=2, disp="smooth")
    assert torch.equal(batch[("disp", 1)], synth.smooth_disp(1, 32, 64, 3, 1))
