"""Static geometry buffers and 4x4 matrix preparation for the lift+render path.

Everything here is init-time or O(B*N) host work; the per-voxel / per-ray
arithmetic lives in the HIP kernels (csrc/).  Values are produced with the same
torch constructors (linspace / arange, fp32) as the reference so that the
buffers -- which are part of every checkpoint -- are bit-identical:

  frustum        /root/reference/src/layers/backbones/base_vampire2.py:253-271
  camera_mids    base_vampire2.py:243-246
  bev_mids       base_vampire2.py:248-251
  voxel coords   base_vampire2.py:273-293
  occ coords     base_vampire2.py:295-312
"""
import torch

from .config import PathConfig, axis_cells


def axis_centres(bound) -> torch.Tensor:
    """Cell centres along one axis: linspace(lo+s/2, hi-s/2, int((hi-lo)/s))."""
    lo, hi, step = bound
    return torch.linspace(lo + step / 2.0, hi - step / 2.0, axis_cells(bound),
                          dtype=torch.float)


def depth_planes(d_bound) -> torch.Tensor:
    return torch.arange(*d_bound, dtype=torch.float)


def make_frustum(final_dim, downsample_factor, d_bound) -> torch.Tensor:
    """[D, fH, fW, 4] homogeneous image-space points (u, v, d, 1)."""
    img_h, img_w = final_dim
    fh, fw = img_h // downsample_factor, img_w // downsample_factor
    d = depth_planes(d_bound)
    u = torch.linspace(0, img_w - 1, fw, dtype=torch.float)
    v = torch.linspace(0, img_h - 1, fh, dtype=torch.float)
    n_d = d.numel()
    out = torch.empty(n_d, fh, fw, 4, dtype=torch.float)
    out[..., 0] = u.view(1, 1, fw)
    out[..., 1] = v.view(1, fh, 1)
    out[..., 2] = d.view(n_d, 1, 1)
    out[..., 3] = 1.0
    return out


def make_camera_mids(d_bound) -> torch.Tensor:
    d = depth_planes(d_bound)
    return 0.5 * (d[:-1] + d[1:])


def make_bev_mids(z_bound_det) -> torch.Tensor:
    """Bin centres of the det-grid height axis, top-down order."""
    return torch.flip(axis_centres(z_bound_det), dims=[0])


def make_voxel_coords(x_bound, y_bound, z_bound, norm: bool = False) -> torch.Tensor:
    """[Z, Y, X, 4] voxel centres (x, y, z, 1); ``norm`` gives [Z, Y, X, 3] in [-1, 1]."""
    zs, ys, xs = axis_centres(z_bound), axis_centres(y_bound), axis_centres(x_bound)
    if norm:
        zs = (zs - z_bound[0]) / (z_bound[1] - z_bound[0])
        ys = (ys - y_bound[0]) / (y_bound[1] - y_bound[0])
        xs = (xs - x_bound[0]) / (x_bound[1] - x_bound[0])
    nz, ny, nx = zs.numel(), ys.numel(), xs.numel()
    gx = xs.view(1, 1, nx).expand(nz, ny, nx)
    gy = ys.view(1, ny, 1).expand(nz, ny, nx)
    gz = zs.view(nz, 1, 1).expand(nz, ny, nx)
    if norm:
        return torch.stack([gx, gy, gz], dim=-1) * 2.0 - 1.0
    return torch.stack([gx, gy, gz, torch.ones_like(gx)], dim=-1)


def make_occ_coords(point_cloud_range=(-40.0, -40.0, -1.0, 40.0, 40.0, 5.4),
                    voxel_size=(0.4, 0.4, 0.4), dims=(200, 200, 16)) -> torch.Tensor:
    """[200, 200, 16, 3] Occ3D voxel centres, indexed [x, y, z]."""
    ix = torch.arange(dims[0]).view(-1, 1, 1).expand(*dims)
    iy = torch.arange(dims[1]).view(1, -1, 1).expand(*dims)
    iz = torch.arange(dims[2]).view(1, 1, -1).expand(*dims)
    cx = ix * voxel_size[0] + voxel_size[0] / 2 + point_cloud_range[0]
    cy = iy * voxel_size[1] + voxel_size[1] / 2 + point_cloud_range[1]
    cz = iz * voxel_size[2] + voxel_size[2] / 2 + point_cloud_range[2]
    return torch.stack([cx, cy, cz], dim=-1).to(torch.float)


# --------------------------------------------------------------------------
# 4x4 matrix preparation (host side of get_pixel / get_geometry)
# --------------------------------------------------------------------------
def lift_matrices(sensor2ego, intrin, ida, bda):
    """Matrices the lift kernel applies, in application order.

    Follows base_vampire2.py:370-387: ``inv(bda)`` (identity when bda is None),
    ``intrin @ inv(sensor2ego)``, ``ida``.  Returns ``[B, N, 3, 4, 4]`` fp32.
    """
    B, N = sensor2ego.shape[:2]
    cam_from_ego = intrin.matmul(torch.inverse(sensor2ego))
    if bda is None:
        inv_bda = torch.eye(4, dtype=sensor2ego.dtype, device=sensor2ego.device).expand(B, 4, 4)
    else:
        inv_bda = torch.inverse(bda)
    inv_bda = inv_bda.unsqueeze(1).expand(B, N, 4, 4)
    return torch.stack([inv_bda, cam_from_ego, ida], dim=2).contiguous().float()


def render_matrices(sensor2ego, intrin, ida, bda):
    """Matrices the camera-render kernel applies to frustum points.

    Follows base_vampire2.py:333-346: ``inv(ida)``, ``sensor2ego @ inv(intrin)``,
    ``bda`` (identity when None).  Returns ``[B, N, 3, 4, 4]`` fp32.
    """
    B, N = sensor2ego.shape[:2]
    inv_ida = torch.inverse(ida)
    ego_from_cam = sensor2ego.matmul(torch.inverse(intrin))
    if bda is None:
        bda_m = torch.eye(4, dtype=sensor2ego.dtype, device=sensor2ego.device).expand(B, 4, 4)
    else:
        bda_m = bda
    bda_m = bda_m.unsqueeze(1).expand(B, N, 4, 4)
    return torch.stack([inv_ida, ego_from_cam, bda_m], dim=2).contiguous().float()


class PathGeometry:
    """All static buffers for one PathConfig (CPU tensors; move with .to())."""

    def __init__(self, cfg: PathConfig):
        self.cfg = cfg
        self.frustum = make_frustum(cfg.final_dim, cfg.downsample_factor, cfg.d_bound)
        self.camera_mids = make_camera_mids(cfg.d_bound)
        self.bev_mids = make_bev_mids(cfg.z_bound_det)
        self.voxel_coords = make_voxel_coords(cfg.x_bound_seg, cfg.y_bound_seg, cfg.z_bound_seg)
        self.output_coords = make_voxel_coords(cfg.x_bound_det, cfg.y_bound_det, cfg.z_bound_det)
