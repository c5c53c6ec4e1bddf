"""Seeded synthetic inputs for the lift+render path (no dataset, no network).

Camera rig follows the nuScenes conventions the reference's loader produces
(/root/reference/src/datasets/nusc_det_seg_dataset.py:118-146 for ``ida``,
:604-659 for sensor2ego / intrinsics) with the numbers of SURVEY.md §8(d):
six pinhole cameras on a ring, optical axis horizontal, image resized/cropped
from 900x1600 to ``final_dim`` in "val" mode.
"""
import math

import torch

from .config import PathConfig

_YAWS_DEG = (55.0, 0.0, -55.0, 110.0, 180.0, -110.0)
_TRANS = ((1.5, 0.5, 1.5), (1.7, 0.0, 1.5), (1.5, -0.5, 1.5),
          (1.0, 0.5, 1.5), (0.0, 0.0, 1.5), (1.0, -0.5, 1.5))


def _rz(deg: float) -> torch.Tensor:
    a = math.radians(deg)
    c, s = math.cos(a), math.sin(a)
    return torch.tensor([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]], dtype=torch.float64)


def camera_rig(cfg: PathConfig, batch: int, *, src_hw=(900, 1600), focal=1266.4,
               centre=(816.3, 491.5), jitter: float = 0.0, seed: int = 0):
    """Return (sensor2ego, intrin, ida) as fp32 ``[B, N, 4, 4]`` tensors.

    ``jitter`` (degrees / metres) perturbs yaw and translation per sample so
    that batch elements do not share geometry.
    """
    g = torch.Generator().manual_seed(seed)
    n_cams = cfg.num_cams
    cam_axes = torch.tensor([[0.0, 0.0, 1.0], [-1.0, 0.0, 0.0], [0.0, -1.0, 0.0]],
                            dtype=torch.float64)
    src_h, src_w = src_hw
    img_h, img_w = cfg.final_dim
    resize = max(img_h / src_h, img_w / src_w)
    new_w, new_h = int(src_w * resize), int(src_h * resize)
    crop_h = new_h - img_h
    crop_w = int(max(0, new_w - img_w) / 2)

    s2e = torch.zeros(batch, n_cams, 4, 4, dtype=torch.float64)
    K = torch.zeros(batch, n_cams, 4, 4, dtype=torch.float64)
    ida = torch.zeros(batch, n_cams, 4, 4, dtype=torch.float64)
    for b in range(batch):
        for n in range(n_cams):
            dyaw, dt = 0.0, torch.zeros(3, dtype=torch.float64)
            if jitter > 0:
                dyaw = float((torch.rand(1, generator=g) - 0.5) * 2 * jitter)
                dt = (torch.rand(3, generator=g, dtype=torch.float64) - 0.5) * 0.1 * jitter
            s2e[b, n, :3, :3] = _rz(_YAWS_DEG[n % 6] + dyaw) @ cam_axes
            s2e[b, n, :3, 3] = torch.tensor(_TRANS[n % 6], dtype=torch.float64) + dt
            s2e[b, n, 3, 3] = 1.0
            K[b, n, 0, 0] = K[b, n, 1, 1] = focal
            K[b, n, 0, 2], K[b, n, 1, 2] = centre
            K[b, n, 2, 2] = K[b, n, 3, 3] = 1.0
            ida[b, n, 0, 0] = ida[b, n, 1, 1] = resize
            ida[b, n, 0, 3], ida[b, n, 1, 3] = -crop_w, -crop_h
            ida[b, n, 2, 2] = ida[b, n, 3, 3] = 1.0
    return s2e.float(), K.float(), ida.float()


def bda_matrix(batch: int, rot_deg: float = 0.0, scale: float = 1.0,
               flip_dx: bool = False, flip_dy: bool = False) -> torch.Tensor:
    """BEV data-augmentation matrix ``[B, 4, 4]`` (rotation about z, scale, flips)."""
    m = torch.eye(4, dtype=torch.float64)
    r = _rz(rot_deg) * scale
    if flip_dx:
        r = torch.diag(torch.tensor([-1.0, 1.0, 1.0], dtype=torch.float64)) @ r
    if flip_dy:
        r = torch.diag(torch.tensor([1.0, -1.0, 1.0], dtype=torch.float64)) @ r
    m[:3, :3] = r
    return m.float().unsqueeze(0).repeat(batch, 1, 1)


def lift_inputs(cfg: PathConfig, batch: int, seed: int = 0, device="cpu", dtype=torch.float32):
    """``depth`` = softmax(randn) over D, ``feat`` = randn."""
    g = torch.Generator().manual_seed(seed)
    depth = torch.randn(batch, cfg.num_cams, cfg.D, cfg.fH, cfg.fW, generator=g).softmax(dim=2)
    feat = torch.randn(batch, cfg.num_cams, cfg.mid_channels, cfg.fH, cfg.fW, generator=g)
    return depth.to(device=device, dtype=dtype), feat.to(device=device, dtype=dtype)


def render_inputs(cfg: PathConfig, batch: int, seed: int = 0, device="cpu", dtype=torch.float32):
    """Synthetic (density_feature, semantic_logits, base, rgb) volumes ``[B, c, Z, Y, X]``."""
    g = torch.Generator().manual_seed(seed + 1000003)
    shp = (cfg.vZ, cfg.vY, cfg.vX)
    density_feature = 0.5 * torch.randn(batch, 1, *shp, generator=g) - 1.0
    semantic_logits = torch.randn(batch, cfg.num_classes, *shp, generator=g)
    base = 0.5 * torch.randn(batch, cfg.mid_channels, *shp, generator=g)
    rgb = torch.rand(batch, 3, *shp, generator=g)
    return tuple(t.to(device=device, dtype=dtype)
                 for t in (density_feature, semantic_logits, base, rgb))
