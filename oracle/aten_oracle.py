"""ORACLE -- test infrastructure only, never shipped, never on the product path.

Unfused torch/aten CPU restatement of the reference's lift + volume-render hot
path.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this file; ``vampire_amd`` must not.

The reference's arithmetic for this path lives in PyTorch aten (pinned
torch==1.9.0 in /root/reference/requirements.txt:11; this image has 2.10.0):
``F.grid_sample`` 5-D bilinear with zero padding, ``torch.inverse``, batched
4x4 ``matmul``, ``exp``/``cumsum``/``sum``/``norm``/``nan_to_num``.  This file
restates the *sequence* of those ops, function by function, each citing the
reference lines it follows (paths relative to /root/reference/).

Parity pin: every function here is checked against outputs of the reference
itself (imported in the build container by ``tests/golden/make_golden.py``);
the resulting vectors are committed under ``tests/golden/`` and verified by
``tests/test_oracle_golden.py``.  The reference ships no tests or golden vectors
of its own, so those generated fixtures are the only pin.
"""
import torch
import torch.nn.functional as F

bv2 = "src/layers/backbones/base_vampire2.py"


# --------------------------------------------------------------------------
# geometry: get_geometry (bv2:314-349) and get_pixel (bv2:351-388)
# --------------------------------------------------------------------------
def frustum_to_ego(frustum, sensor2ego, intrin, ida, bda, prepared=None):
    """bv2:328-349.  frustum [D,fH,fW,4] -> ego points [B,N,D,fH,fW,3].

    ``prepared`` [B,N,3,4,4] = (inv(ida), sensor2ego @ inv(intrin), bda) replaces the three
    host-side 4x4 products: ``torch.inverse`` is LAPACK and its last bits depend on the CPU,
    so bit-exact comparisons across machines pin the prepared matrices."""
    B, N = (sensor2ego if prepared is None else prepared).shape[:2]
    if prepared is None:
        inv_ida = ida.view(B, N, 1, 1, 1, 4, 4).inverse()
        ego_from_cam = sensor2ego.matmul(torch.inverse(intrin)).view(B, N, 1, 1, 1, 4, 4)
        bda_m = None if bda is None else bda.unsqueeze(1).repeat(1, N, 1, 1).view(B, N, 1, 1, 1, 4, 4)
    else:
        inv_ida, ego_from_cam, bda_m = (prepared[:, :, i].reshape(B, N, 1, 1, 1, 4, 4) for i in range(3))
    pts = inv_ida.matmul(frustum.unsqueeze(-1))
    # (u*d, v*d, d, 1): undo the perspective divide (bv2:336-338)
    pts = torch.cat((pts[..., :2, :] * pts[..., 2:3, :], pts[..., 2:, :]), dim=5)
    pts = ego_from_cam.matmul(pts)
    if bda_m is not None:
        pts = bda_m @ pts
    return pts.squeeze(-1)[..., :3]


def ego_to_pixel(voxel_coords, sensor2ego, intrin, ida, bda, prepared=None):
    """bv2:365-388.  voxel centres [Z,Y,X,4] -> (u, v, depth) [B,N,Z,Y,X,3].

    ``prepared`` [B,N,3,4,4] = (inv(bda), intrin @ inv(sensor2ego), ida), see frustum_to_ego."""
    B, N = (sensor2ego if prepared is None else prepared).shape[:2]
    pts = voxel_coords.unsqueeze(-1)
    if prepared is None:
        inv_bda = None if bda is None else \
            bda.unsqueeze(1).repeat(1, N, 1, 1).view(B, N, 1, 1, 1, 4, 4).inverse()
        cam_from_ego = intrin.matmul(torch.inverse(sensor2ego)).view(B, N, 1, 1, 1, 4, 4)
        ida_m = ida.view(B, N, 1, 1, 1, 4, 4)
    else:
        inv_bda, cam_from_ego, ida_m = (prepared[:, :, i].reshape(B, N, 1, 1, 1, 4, 4) for i in range(3))
    if inv_bda is not None:
        pts = inv_bda.matmul(pts)
    pts = cam_from_ego.matmul(pts)
    zc = torch.clamp(pts[..., 2:3, :], min=1e-6)           # bv2:383-385
    pts = torch.cat((pts[..., :2, :] / zc, pts[..., 2:, :]), dim=5)
    pts = ida_m.matmul(pts).squeeze(-1)
    return pts[..., :3]


# --------------------------------------------------------------------------
# LIFT: get_voxel_feats (bv2:483-516; D==1 variant base_bilinear.py:471-519)
# --------------------------------------------------------------------------
def lift_valid_and_grid(pix, final_dim, d_bound, use_depth=True, clamp_extreme=True):
    """bv2:493-506.  Returns (valid [B,N,Z,Y,X] float, norm grid [B,N,Z,Y,X,3])."""
    u, v, z = pix[..., 0], pix[..., 1], pix[..., 2]
    ok = (u > -0.5) & (u < float(final_dim[1] - 0.5)) & \
         (v > -0.5) & (v < float(final_dim[0] - 0.5))
    if use_depth:
        ok = ok & (z > d_bound[0]) & (z < d_bound[1])
    else:
        ok = ok & (z > 0.0)                                   # base_bilinear.py:486
    nx = 2.0 * (u / float(final_dim[1] - 1)) - 1.0
    ny = 2.0 * (v / float(final_dim[0] - 1)) - 1.0
    if use_depth:
        nz = 2.0 * ((z - d_bound[0]) / (d_bound[1] - d_bound[0])) - 1.0
    else:
        nz = torch.zeros_like(nx)                             # base_bilinear.py:510
    if clamp_extreme:
        nx, ny = nx.clamp(-2.0, 2.0), ny.clamp(-2.0, 2.0)
        if use_depth:
            nz = nz.clamp(-2.0, 2.0)
    return ok.float(), torch.stack([nx, ny, nz], dim=-1)


def lift_from_frustum_feats(frustum_feats, pix, final_dim, d_bound, use_depth=True):
    """bv2:507-516.  frustum_feats [B,N,C,D,fH,fW] -> voxel mean [B,C,Z,Y,X]."""
    B, N, C, D, H, W = frustum_feats.shape
    Z, Y, X = pix.shape[2:5]
    valid, grid = lift_valid_and_grid(pix, final_dim, d_bound, use_depth)
    s = F.grid_sample(frustum_feats.reshape(B * N, C, D, H, W),
                      grid.reshape(B * N, Z, Y, X, 3), align_corners=False)
    s = s.reshape(B, N, C, Z, Y, X) * valid.unsqueeze(2)
    hit = (s.abs() > 0).float()
    return s.sum(dim=1) / (hit.sum(dim=1) + 1e-6)


def outer_depth_feat(depth, feat):
    """bv2:553.  depth [B,N,D,h,w] (x) feat [B,N,C,h,w] -> [B,N,C,D,h,w]."""
    return depth.unsqueeze(2) * feat.unsqueeze(3)


def lift(depth, feat, voxel_coords, sensor2ego, intrin, ida, bda, final_dim, d_bound,
         prepared=None):
    """bv2:550-563 end to end (outer product + get_pixel + get_voxel_feats)."""
    pix = ego_to_pixel(voxel_coords, sensor2ego, intrin, ida, bda, prepared)
    return lift_from_frustum_feats(outer_depth_feat(depth, feat), pix, final_dim, d_bound)


def lift_tap_indices(pix, final_dim, d_bound, frustum_shape):
    """Floor-corner tap indices of the lift's grid_sample (align_corners=False).

    aten unnormalises as ``((g + 1) * size - 1) / 2`` (GridSampler.h,
    grid_sampler_unnormalize); the eight taps are floor(.) and floor(.)+1.
    Returns (valid bool, ix0, iy0, iz0 int32), each [B,N,Z,Y,X].
    """
    D, H, W = frustum_shape
    valid, grid = lift_valid_and_grid(pix, final_dim, d_bound)
    fx = ((grid[..., 0] + 1) * W - 1) / 2
    fy = ((grid[..., 1] + 1) * H - 1) / 2
    fz = ((grid[..., 2] + 1) * D - 1) / 2
    to_i = lambda t: torch.floor(t).to(torch.int32)
    return valid.bool(), to_i(fx), to_i(fy), to_i(fz)


# --------------------------------------------------------------------------
# density activations (src/utils/render_utils.py:30-46; nn.Sigmoid for 'naive',
# bv2:191-194)
# --------------------------------------------------------------------------
def density_sdf(s, beta_param, bias, beta_min=1e-4):
    beta = beta_param.abs() + beta_min                       # render_utils.py:44-46
    t = s - bias
    return (1 / beta) * (0.5 + 0.5 * t.sign() * torch.expm1(-t.abs() / beta))


def density_apply(s, mode, beta_param=None, bias=-1.0):
    return torch.sigmoid(s) if mode == "naive" else density_sdf(s, beta_param, bias)


# --------------------------------------------------------------------------
# RENDER: volume_rendering_from_multiple_views (bv2:391-467)
# --------------------------------------------------------------------------
def _composite(sigma_delta, dim):
    """alpha * exp(-exclusive_cumsum) along ``dim`` (bv2:430-434 / 454-458)."""
    alpha = 1 - torch.exp(-sigma_delta)
    n = sigma_delta.shape[dim]
    head = torch.zeros_like(sigma_delta.narrow(dim, 0, 1))
    excl = torch.cat([head, torch.cumsum(sigma_delta.narrow(dim, 0, n - 1), dim=dim)], dim=dim)
    return alpha * torch.exp(-excl)


def render(geom, density_feature, semantic_logits, base, rgb, *, seg_bounds, output_coords,
           camera_mids, bev_mids, d_far, z_step_det, num_classes, density_mode,
           beta_param=None, sdf_bias=-1.0, cat_seg=False):
    """bv2:391-467.  Returns the reference's 8-tuple.

    geom [B,N,D,fH,fW,3] (already nan_to_num'ed with -1e3 as at bv2:612);
    volumes [B,c,Z,Y,X]; seg_bounds = (x_bound_seg, y_bound_seg, z_bound_seg).
    """
    B, N, D, H, W, _ = geom.shape
    K = num_classes
    vol = torch.cat([density_feature, semantic_logits, rgb, base], dim=1)   # bv2:396
    lo = torch.tensor([b[0] for b in seg_bounds], dtype=geom.dtype)
    span = torch.tensor([b[1] - b[0] for b in seg_bounds], dtype=geom.dtype)

    # ---- camera branch (bv2:397-440) ----
    g = (geom[:, :, :-1] - lo) / span
    g = g * 2.0 - 1.0
    inside = ((g >= -1.0) & (g <= 1.0)).all(dim=-1)
    S = F.grid_sample(vol, g.reshape(B, -1, H, W, 3), align_corners=True)
    S = S.reshape(B, -1, N, D - 1, H, W).permute(0, 2, 1, 3, 4, 5) * inside.unsqueeze(2)
    S = torch.nan_to_num(S)
    sigma = density_apply(S[:, :, :1], density_mode, beta_param, sdf_bias)
    delta = torch.norm(geom[:, :, 1:] - geom[:, :, :-1], dim=-1)
    wts = _composite(sigma * delta.unsqueeze(2), dim=3)
    acc = wts.sum(dim=3)
    rgb_preds = (wts * S[:, :, K + 1:K + 4]).sum(dim=3)
    seg_preds = (wts * S[:, :, 1:K + 1]).sum(dim=3)
    depth_preds = (wts * camera_mids[None, None, None, :, None, None]).sum(dim=3) \
        + (1 - acc) * d_far                                                 # bv2:436,440

    # ---- BEV branch (bv2:408-418, 442-461) ----
    og = (output_coords[..., :3] - lo) / span
    og = (og * 2.0 - 1.0)[None].expand(B, *og.shape)
    Vs = torch.flip(F.grid_sample(vol, og, align_corners=True), dims=[2])
    v_sigma = density_apply(Vs[:, :1], density_mode, beta_param, sdf_bias)
    v_out = Vs[:, K + 4:]
    if cat_seg:
        v_out = torch.cat((v_out, Vs[:, 1:K + 1]), dim=1)
    v_w = _composite(v_sigma * (torch.ones_like(v_sigma) * z_step_det), dim=2)
    bev_rgb = (v_w * Vs[:, K + 1:K + 4]).sum(dim=2)
    bev_seg = (v_w * Vs[:, 1:K + 1]).sum(dim=2)
    bev_height = (v_w * bev_mids[None, None, :, None, None]).sum(dim=2)
    return rgb_preds, seg_preds, depth_preds, bev_rgb, bev_seg, bev_height, v_sigma, v_out


def render_tap_indices(geom, seg_bounds, vol_shape):
    """Floor taps + inside mask of the camera branch's grid_sample (align_corners=True).

    aten unnormalises as ``(g + 1) / 2 * (size - 1)``.
    """
    Z, Y, X = vol_shape
    lo = torch.tensor([b[0] for b in seg_bounds], dtype=geom.dtype)
    span = torch.tensor([b[1] - b[0] for b in seg_bounds], dtype=geom.dtype)
    g = ((geom[:, :, :-1] - lo) / span) * 2.0 - 1.0
    inside = ((g >= -1.0) & (g <= 1.0)).all(dim=-1)
    fx = (g[..., 0] + 1) / 2 * (X - 1)
    fy = (g[..., 1] + 1) / 2 * (Y - 1)
    fz = (g[..., 2] + 1) / 2 * (Z - 1)
    to_i = lambda t: torch.floor(t).to(torch.int32)
    return inside, to_i(fx), to_i(fy), to_i(fz)


# --------------------------------------------------------------------------
# point resampling: occupancy grid and lidar points (bv2:576-609)
# --------------------------------------------------------------------------
def normalise_points(points, seg_bounds):
    """(p - lo) / (hi - lo) * 2 - 1 per axis (bv2:581-586, 599-602); points [..., 3] ego xyz."""
    lo = torch.as_tensor([b[0] for b in seg_bounds], dtype=points.dtype)
    span = torch.as_tensor([b[1] - b[0] for b in seg_bounds], dtype=points.dtype)
    return (points - lo) / span * 2.0 - 1.0


def sample_points(volume, points, seg_bounds, padding="zeros", mask_outside=False):
    """volume [B,C,Z,Y,X], points [B,P,3] ego xyz -> [B,C,P]: the reference's
    F.grid_sample(volume, norm_pts, padding_mode=..., align_corners=True) (bv2:590, 594, 603-604),
    optionally times the all(-1 <= n <= 1) mask (bv2:587-589, 595)."""
    n = normalise_points(points, seg_bounds)
    out = F.grid_sample(volume, n[:, None, None], padding_mode=padding, align_corners=True)[:, :, 0, 0]
    if mask_outside:
        out = out * ((n >= -1.0) & (n <= 1.0)).all(dim=-1).unsqueeze(1)
    return out


def occupancy_queries(semantic_logits, density_feature, occ_coords, bda_mat, seg_bounds, density_mode,
                      beta_param=None, sdf_bias=-1.0):
    """bv2:596-604: the fixed occ grid rotated by bda[:3,:3], then occ_logits (border padding) and
    occ_density (the activated density, zero padding); both [B,c,oz,oy,ox]."""
    B = semantic_logits.shape[0]
    rot = bda_mat[:, :3, :3].view(B, 1, 1, 1, 3, 3)
    occ = (rot @ occ_coords[None, ..., None].expand(B, *occ_coords.shape, 1)).squeeze(-1)
    pts = occ.reshape(B, -1, 3)
    shp = occ_coords.shape[:3]
    logits = sample_points(semantic_logits, pts, seg_bounds, "border").reshape(B, -1, *shp)
    dens = sample_points(density_apply(density_feature, density_mode, beta_param, sdf_bias), pts,
                         seg_bounds, "zeros").reshape(B, 1, *shp)
    return logits, dens


# --------------------------------------------------------------------------
# whole hot path, for the timed CPU baseline (bench.py cpu_baseline)

# --------------------------------------------------------------------------
# producer / consumer glue either side of the path (bv2:550, 627-630)
# --------------------------------------------------------------------------
def depth_softmax(logits):
    """`mapping_along_depth(src).softmax(dim=1)` (bv2:550): logits [B*N, D, fH, fW]."""
    return logits.float().softmax(dim=1)


def density_gate(voxel_output, voxel_density, density_mode):
    """bv2:627-630: `voxel_output * bev_density.tanh()` for the sdf density, `* bev_density` else."""
    return voxel_output * (voxel_density.tanh() if density_mode == "sdf" else voxel_density)


def upsample_trilinear(x, size):
    """The resize inside Hourglass3D (bv2:66, 72): F.interpolate(..., mode='trilinear',
    align_corners=True)."""
    return F.interpolate(x, tuple(size), mode="trilinear", align_corners=True)


# --------------------------------------------------------------------------
def lift_render_forward(cfg, geo, depth, feat, vols, mats, beta_param, prepared=(None, None)):
    """One lift + render forward at the reference's op sequence.  ``mats`` =
    (sensor2ego, intrin, ida, bda); ``vols`` = (density_feature, sem, base, rgb);
    ``prepared`` = optional (lift_mats, render_mats)."""
    s2e, K, ida, bda = mats
    vox = lift(depth, feat, geo.voxel_coords, s2e, K, ida, bda, cfg.final_dim, cfg.d_bound,
               prepared[0])
    geom = torch.nan_to_num(frustum_to_ego(geo.frustum, s2e, K, ida, bda, prepared[1]), -1e3)  # bv2:612
    outs = render(geom, *vols,
                  seg_bounds=(cfg.x_bound_seg, cfg.y_bound_seg, cfg.z_bound_seg),
                  output_coords=geo.output_coords, camera_mids=geo.camera_mids,
                  bev_mids=geo.bev_mids, d_far=cfg.d_bound[1], z_step_det=cfg.z_bound_det[2],
                  num_classes=cfg.num_classes, density_mode=cfg.density_mode,
                  beta_param=beta_param, sdf_bias=cfg.sdf_bias, cat_seg=cfg.cat_seg)
    return vox, outs
