"""CPU oracle for the view-synthesis loss path.  TEST INFRASTRUCTURE ONLY.

This file restates, in plain PyTorch ops, what the reference computes on the
path  disp -> depth -> back-projection -> projection -> bilinear warp ->
SSIM/L1 photometric error -> auto-mask / per-pixel minimum -> edge-aware
smoothness -> per-scale loss assembly.  Nothing in the shipped package imports
it: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` do, and there it is the checker, never the thing measured as the
product.

Parity status: PINNED.  ``oracle/make_golden.py`` runs the reference's own
``layers.py`` / ``Trainer.py`` / ``Trainer_distilation.py`` (loaded by file
path in the development container) on seeded inputs and stores inputs, every
intermediate, the loss and the gradients under ``tests/golden``;
``tests/test_oracle_golden.py`` checks this restatement against those files.

Reference lines followed (all under /root/reference/Python):
  disp_to_depth ............ layers.py:16-25
  back-projection .......... layers.py:139-174
  projection ............... layers.py:177-202
  warp ..................... Trainer.py:467 (F.grid_sample, border padding)
  SSIM ..................... layers.py:229-262
  photometric error ........ Trainer.py:481-495
  auto-mask / minimum ...... Trainer.py:551-605
  smoothness ............... layers.py:211-226, Trainer.py:609-614
  loss assembly ............ Trainer.py:524-536,614-619
  0.9 photometric scale .... Trainer_distilation.py:1232-1234
"""
from __future__ import annotations

import dataclasses
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

SSIM_C1 = 0.01 ** 2
SSIM_C2 = 0.03 ** 2


# --------------------------------------------------------------------------
# configuration
# --------------------------------------------------------------------------
@dataclasses.dataclass
class LossConfig:
    """The option fields the path reads (options.py:40-72,105-135)."""
    frame_ids: Sequence = (0, -1, 1)
    scales: Sequence[int] = (0,)
    height: int = 192
    width: int = 640
    min_depth: float = 0.1
    max_depth: float = 100.0
    automask: bool = True               # == not opt.disable_automasking
    avg_reprojection: bool = False
    no_ssim: bool = False
    disparity_smoothness: float = 1e-3
    v1_multiscale: bool = False
    warp_scale: float = 1.0             # 0.9 in Trainer_distilation.compute_losses
    num_scales: int = 1                 # Trainer.py:33 (divisor of the total)
    pose_model_type: str = "shared"     # "posecnn": T rebuilt from the parameters x mean inverse depth
    use_affinity: bool = False          # Trainer.py:26 (and not val_mode): affinity warp + loss

    @property
    def sources(self) -> List:
        return list(self.frame_ids[1:])


# --------------------------------------------------------------------------
# the three tiny contractions of the path
# --------------------------------------------------------------------------
# The reference calls torch.matmul for K@T, inv_K@pix and P@points (layers.py:168,196-198).
# In float32 the last bits of a BLAS product depend on the host's BLAS kernels (summation
# order, use of fused multiply-add), and those bits move the sampling coordinates enough to
# change float32 gradients at the 1e-5 level.  To give the same numbers on every host the
# oracle spells the contractions out in the order and rounding the reference produced on the
# host that generated tests/golden (bit-for-bit, see tests/test_oracle_golden.py):
#   [B,4,4] @ [B,4,4]  : products and sums rounded separately, k ascending;
#   [B,m,k] @ [B,k,HW] : first product rounded, then one fused multiply-add per further k.
# FIXED_ORDER = False switches back to torch.matmul.
FIXED_ORDER = True


def _fma32(a: torch.Tensor, b: torch.Tensor, c: torch.Tensor) -> torch.Tensor:
    """float32 fused multiply-add: the product of two float32 is exact in float64."""
    return (a.double() * b.double() + c.double()).to(torch.float32)


def _contract(A: torch.Tensor, X: torch.Tensor, fused: bool) -> torch.Tensor:
    """A [B,m,k] @ X [B,k,n] (k <= 4) in a fixed order; other dtypes go to torch.matmul."""
    if not FIXED_ORDER or A.dtype != torch.float32 or X.dtype != torch.float32:
        return torch.matmul(A, X)
    k = A.shape[2]
    acc = A[:, :, 0:1] * X[:, 0:1, :]
    for i in range(1, k):
        a, x = A[:, :, i:i + 1], X[:, i:i + 1, :]
        acc = _fma32(a, x, acc) if fused else acc + a * x
    return acc


# --------------------------------------------------------------------------
# geometry
# --------------------------------------------------------------------------
def disp_to_depth(disp: torch.Tensor, min_depth: float, max_depth: float):
    """layers.py:16-25 -- sigmoid disparity to (scaled disparity, depth)."""
    lo = 1 / max_depth
    hi = 1 / min_depth
    scaled = lo + (hi - lo) * disp
    return scaled, 1 / scaled


def homogeneous_pixels(height: int, width: int, device=None, dtype=torch.float32) -> torch.Tensor:
    """layers.py:149-159 -- [3, H*W] rows (x, y, 1), x fastest; float32 in the reference
    (other dtypes only for high-precision evaluations of the same formulas)."""
    ys, xs = torch.meshgrid(
        torch.arange(height, dtype=dtype, device=device),
        torch.arange(width, dtype=dtype, device=device),
        indexing="ij",
    )
    return torch.stack([xs.reshape(-1), ys.reshape(-1), torch.ones(height * width, dtype=dtype, device=device)], 0)


def backproject(depth: torch.Tensor, inv_K: torch.Tensor) -> torch.Tensor:
    """layers.py:161-174 -- depth [B,1,H,W], inv_K [B,4,4] -> points [B,4,H*W]."""
    b, _, h, w = depth.shape
    pix = homogeneous_pixels(h, w, depth.device, depth.dtype).unsqueeze(0).expand(b, 3, h * w)
    rays = _contract(inv_K[:, :3, :3], pix, fused=True)
    pts = depth.reshape(b, 1, -1) * rays
    ones = torch.ones(b, 1, h * w, dtype=pts.dtype, device=pts.device)
    return torch.cat([pts, ones], 1)


def project(points: torch.Tensor, K: torch.Tensor, T: torch.Tensor,
            height: int, width: int, eps: float = 1e-7) -> torch.Tensor:
    """layers.py:188-202 -- points [B,4,HW] -> normalised grid [B,H,W,2]."""
    b = points.shape[0]
    P = _contract(K, T, fused=False)[:, :3, :]
    cam = _contract(P, points, fused=True)
    uv = cam[:, :2, :] / (cam[:, 2, :].unsqueeze(1) + eps)
    uv = uv.reshape(b, 2, height, width).permute(0, 2, 3, 1)
    scale = torch.tensor([width - 1, height - 1], dtype=uv.dtype, device=uv.device)
    uv = uv / scale
    return (uv - 0.5) * 2


def warp(image: torch.Tensor, grid: torch.Tensor) -> torch.Tensor:
    """Trainer.py:467 -- bilinear, border padding, align_corners=False; the grid is
    promoted to the image dtype (float64 in the reference's data path)."""
    return F.grid_sample(image, grid.to(image.dtype), mode="bilinear",
                         padding_mode="border", align_corners=False)


def warp_features(feats: torch.Tensor, disp: torch.Tensor, K: torch.Tensor, inv_K: torch.Tensor, T: torch.Tensor,
                  min_depth: float = 0.1, max_depth: float = 100.0):
    """Train_distillation_new.py:1159-1164 (and :1168-1182 at the coarser sizes): disparity already resized to
    the feature map's size -> depth -> points -> grid -> grid_sample of the float32 features.  Returns
    (warped features, grid)."""
    _, depth = disp_to_depth(disp, min_depth, max_depth)
    h, w = feats.shape[2:]
    grid = project(backproject(depth, inv_K), K, T, h, w)
    return warp(feats, grid), grid


# --------------------------------------------------------------------------
# photometric terms
# --------------------------------------------------------------------------
def ssim(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """layers.py:245-262 -- per-pixel (1-SSIM)/2 over a reflect-padded 3x3 window."""
    x = F.pad(x, (1, 1, 1, 1), mode="reflect")
    y = F.pad(y, (1, 1, 1, 1), mode="reflect")

    def box(t):
        return F.avg_pool2d(t, 3, 1)

    mu_x, mu_y = box(x), box(y)
    var_x = box(x ** 2) - mu_x ** 2
    var_y = box(y ** 2) - mu_y ** 2
    cov = box(x * y) - mu_x * mu_y
    num = (2 * mu_x * mu_y + SSIM_C1) * (2 * cov + SSIM_C2)
    den = (mu_x ** 2 + mu_y ** 2 + SSIM_C1) * (var_x + var_y + SSIM_C2)
    return torch.clamp((1 - num / den) / 2, 0, 1)


def reprojection_error(pred: torch.Tensor, target: torch.Tensor, no_ssim: bool = False) -> torch.Tensor:
    """Trainer.py:481-495 -- [B,1,H,W] photometric error."""
    l1 = torch.abs(target - pred).mean(1, True)
    if no_ssim:
        return l1
    return 0.85 * ssim(pred, target).mean(1, True) + 0.15 * l1


def smoothness(disp: torch.Tensor, img: torch.Tensor) -> torch.Tensor:
    """layers.py:211-226 -- edge-aware first-order smoothness (0-dim)."""
    ddx = torch.abs(disp[:, :, :, :-1] - disp[:, :, :, 1:])
    ddy = torch.abs(disp[:, :, :-1, :] - disp[:, :, 1:, :])
    idx = torch.abs(img[:, :, :, :-1] - img[:, :, :, 1:]).mean(1, keepdim=True)
    idy = torch.abs(img[:, :, :-1, :] - img[:, :, 1:, :]).mean(1, keepdim=True)
    # the reference multiplies in place, so the product keeps disp's dtype
    ddx = (ddx * torch.exp(-idx)).to(disp.dtype)
    ddy = (ddy * torch.exp(-idy)).to(disp.dtype)
    return ddx.mean() + ddy.mean()


# --------------------------------------------------------------------------
# trainer-level steps
# --------------------------------------------------------------------------
def view_synthesis(inputs: Dict, outputs: Dict, cfg: LossConfig) -> None:
    """Trainer.py:419-479 -- fills ("depth",0), ("sample",f), ("color",f)."""
    for _scale in cfg.scales:
        disp = outputs[("disp", 0)]
        if not cfg.v1_multiscale:
            disp = F.interpolate(disp, [cfg.height, cfg.width], mode="bilinear", align_corners=False)
        _, depth = disp_to_depth(disp, cfg.min_depth, cfg.max_depth)
        outputs[("depth", 0)] = depth
        for f in cfg.sources:
            T = inputs["stereo_T"] if f == "s" else outputs[("cam_T_cam", f)]
            if cfg.pose_model_type == "posecnn":                    # Trainer.py:446-455
                mean_inv_depth = (1 / depth).mean(3, True).mean(2, True)
                T = transformation_from_parameters(outputs[("axisangle", f)][:, 0],
                                                   outputs[("translation", f)][:, 0] * mean_inv_depth[:, 0], f < 0)
            pts = backproject(depth, inputs["inv_K"])
            grid = project(pts, inputs["K"], T, cfg.height, cfg.width)
            outputs[("sample", f)] = grid
            outputs[("color", f)] = warp(inputs[("image", f)], grid)
            if cfg.use_affinity:                                     # Trainer.py:468-473
                outputs[("pred_affinity", f)] = warp(inputs[("affinity", f)][:, 0].double(), grid)
            if cfg.automask:
                outputs[("color_identity", f)] = inputs[("image", f)]


def affinity_reprojection_loss(pred: torch.Tensor, target: torch.Tensor, weight: Sequence[float],
                               rand: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Trainer.py:497-516 -- squared error between the warped affinity map and the target frame's map, the target
    multiplied by the prediction itself where ``rand < weight[1]`` (the segmentation sums of :508-510 do not enter
    the value).  ``rand`` is the U[0,1) tensor of :502, drawn with torch.rand_like when omitted."""
    rand = torch.rand_like(target) if rand is None else rand
    keep = (rand < weight[1]).float()
    drop = torch.where(keep == 1.0, pred, torch.ones_like(target))
    return torch.mean((pred - target * drop) ** 2)


def affinity_huber_loss(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """Trainer_distilation.py:1076-1110 -- the distillation trainer's variant: Huber loss (delta 1, mean over all
    elements) between the warped map and the target frame's, times the fraction of pixels whose 25 target channels
    sum to less than 24 (:1081-1084; the scalar Huber value is broadcast over that [B,1,H,W] indicator and the
    product averaged, :1104-1110).  ``t_seg`` and ``weight`` of the reference's signature do not enter the value."""
    d = (pred - target.to(pred.dtype)).abs()
    huber = torch.where(d < 1.0, 0.5 * d * d, d - 0.5).mean()
    frac = (target.sum(1, keepdim=True) < 24).to(pred.dtype).mean()
    return huber * frac


def compute_losses(inputs: Dict, outputs: Dict, cfg: LossConfig,
                   noise: Optional[torch.Tensor] = None, drop_rand: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
    """Trainer.py:518-620 (predictive-mask branch excluded).

    ``noise`` is the N(0,1) tie-break tensor of Trainer.py:589 ([B,S,H,W], or
    [B,1,H,W] with avg_reprojection); drawn from torch.randn when omitted.
    ``drop_rand`` pins the uniform tensor of the affinity term (Trainer.py:502).
    """
    losses: Dict[str, torch.Tensor] = {}
    total = 0
    target = inputs[("image", 0)]
    for scale in cfg.scales:
        disp = outputs[("disp", 0)]
        warped = torch.cat(
            [reprojection_error(outputs[("color", f)], target, cfg.no_ssim) * cfg.warp_scale
             if cfg.warp_scale != 1.0 else
             reprojection_error(outputs[("color", f)], target, cfg.no_ssim)
             for f in cfg.sources], 1)
        if cfg.avg_reprojection:
            warped = warped.mean(1, keepdim=True)

        if cfg.automask:
            ident = torch.cat(
                [reprojection_error(inputs[("image", f)], target, cfg.no_ssim) for f in cfg.sources], 1)
            if cfg.avg_reprojection:
                ident = ident.mean(1, keepdim=True)
            z = torch.randn(ident.shape) if noise is None else noise
            ident = ident + z.to(ident.device) * 0.00001
            combined = torch.cat((ident, warped), dim=1)
        else:
            combined = warped

        if combined.shape[1] == 1:
            chosen = combined
        else:
            chosen, idx = torch.min(combined, dim=1)
        if cfg.automask:
            outputs["identity_selection"] = (idx > ident.shape[1] - 1).float()

        loss = chosen.mean()
        if cfg.use_affinity:                                         # Trainer.py:546-548,606
            for f in cfg.sources:
                loss = loss + affinity_reprojection_loss(outputs[("pred_affinity", f)], inputs[("affinity", 0)][:, 0],
                                                         [1.5, 0.5], drop_rand)
        mean_disp = disp.mean(2, True).mean(3, True)
        loss = loss + cfg.disparity_smoothness * smoothness(disp / (mean_disp + 1e-7), target) / (2 ** scale)
        total = total + loss
        losses["loss/{}".format(scale)] = loss
    losses["loss"] = total / cfg.num_scales
    return losses


def loss_forward_backward(inputs: Dict, outputs: Dict, cfg: LossConfig,
                          noise: Optional[torch.Tensor] = None, drop_rand: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
    """One pass of the whole path: view synthesis, losses, backward."""
    view_synthesis(inputs, outputs, cfg)
    losses = compute_losses(inputs, outputs, cfg, noise, drop_rand)
    losses["loss"].backward()
    return losses


# --------------------------------------------------------------------------
# pose helpers (layers.py:28-103) -- tiny [B,4,4] math used to build inputs
# --------------------------------------------------------------------------
def rotation_from_axisangle(vec: torch.Tensor) -> torch.Tensor:
    """layers.py:64-103 -- vec [B,1,3] -> [B,4,4] (Rodrigues)."""
    angle = torch.norm(vec, 2, 2, True)
    axis = vec / (angle + 1e-7)
    ca, sa = torch.cos(angle), torch.sin(angle)
    C = 1 - ca
    x, y, z = (axis[..., i].unsqueeze(1) for i in range(3))
    # association order as in the reference: a*(b*C), so results are bit-identical
    xy, yz, zx = x * (y * C), y * (z * C), z * (x * C)
    rows = [
        [x * (x * C) + ca, xy - z * sa, zx + y * sa],
        [xy + z * sa, y * (y * C) + ca, yz - x * sa],
        [zx - y * sa, yz + x * sa, z * (z * C) + ca],
    ]
    R = torch.zeros(vec.shape[0], 4, 4, dtype=vec.dtype, device=vec.device)
    for i in range(3):
        for j in range(3):
            R[:, i, j] = rows[i][j].reshape(-1)
    R[:, 3, 3] = 1
    return R


def transformation_from_parameters(axisangle: torch.Tensor, translation: torch.Tensor,
                                   invert: bool = False) -> torch.Tensor:
    """layers.py:28-61 -- (axisangle [B,1,3], translation [B,1,3]) -> [B,4,4]."""
    R = rotation_from_axisangle(axisangle)
    t = translation.clone()
    if invert:
        R = R.transpose(1, 2)
        t = t * -1
    Tm = torch.eye(4, dtype=t.dtype, device=t.device).repeat(t.shape[0], 1, 1)
    Tm[:, :3, 3] = t.reshape(-1, 3)
    return torch.matmul(R, Tm) if invert else torch.matmul(Tm, R)


# --------------------------------------------------------------------------
# depth metrics (Trainer.py:622-662, layers.py:265-283)
# --------------------------------------------------------------------------
def depth_errors(gt: torch.Tensor, pred: torch.Tensor):
    """layers.py:265-283 -- (abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3) of two 1-D tensors."""
    thresh = torch.max(gt / pred, pred / gt)
    a1 = (thresh < 1.25).float().mean()
    a2 = (thresh < 1.25 ** 2).float().mean()
    a3 = (thresh < 1.25 ** 3).float().mean()
    rmse = torch.sqrt(((gt - pred) ** 2).mean())
    rmse_log = torch.sqrt(((torch.log(gt) - torch.log(pred)) ** 2).mean())
    abs_rel = torch.mean(torch.abs(gt - pred) / gt)
    sq_rel = torch.mean((gt - pred) ** 2 / gt)
    return abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3


def depth_losses(depth: torch.Tensor, depth_gt: torch.Tensor, crop=(153, 371, 44, 1197)):
    """Trainer.py:622-662 -- depth [B,1,H,W] resized to depth_gt's size (375 x 1242 in the reference), clamped,
    masked by gt > 0 inside the crop, median-scaled over the batch, clamped; returns (sum of the metrics, metrics)."""
    pred = torch.clamp(F.interpolate(depth, list(depth_gt.shape[2:]), mode="bilinear", align_corners=False), 1e-3, 80).detach()
    mask = depth_gt > 0
    crop_mask = torch.zeros_like(mask)
    crop_mask[:, :, crop[0]:crop[1], crop[2]:crop[3]] = 1
    mask = mask * crop_mask
    gt, pred = depth_gt[mask], pred[mask]
    pred = pred * (torch.median(gt) / torch.median(pred))
    pred = torch.clamp(pred, min=1e-3, max=80)
    m = depth_errors(gt, pred)
    return sum(m), list(m)
