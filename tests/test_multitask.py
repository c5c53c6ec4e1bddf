"""SURVEY 8f N4 / BASELINE configs[4]: the multi-task model around the hot path -- CPU tests (the hot-path
operators are replaced by the oracle here; the GPU run of the same step is in test_hip_parity.py)."""
import dataclasses
import math
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from vampire_amd import multitask as M                      # noqa: E402
from vampire_amd.config import CFG_TINY                     # noqa: E402


# ----------------------------------------------------------------------------- losses
@pytest.mark.parametrize("case", ["small", "present_subset", "one_class", "single_pixel", "large"])
def test_lovasz_softmax_matches_reference_golden(case):
    """Vectors made by tests/golden/make_lovasz_golden.py from the reference's lovasz_losses.py:153-199."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "lovasz_golden.npz"))
    x = torch.from_numpy(g[case + "_logits"]).requires_grad_(True)
    labels = torch.from_numpy(g[case + "_labels"])
    loss = M.lovasz_softmax(F.softmax(x, dim=1), labels)
    assert float(loss) == pytest.approx(float(g[case + "_loss"]), rel=2e-6, abs=1e-7)
    loss.backward()
    ref = torch.from_numpy(g[case + "_grad"])
    assert float((x.grad - ref).abs().max()) <= 1e-7 + 2e-5 * float(ref.abs().max())


def test_lovasz_empty_and_ce_lovasz():
    assert float(M.lovasz_softmax(torch.zeros(0, 5), torch.zeros(0, dtype=torch.long))) == 0.0
    torch.manual_seed(0)
    lg, lb = torch.randn(50, 7), torch.randint(0, 7, (50,))
    assert float(M._ce_lovasz(lg, lb)) == pytest.approx(
        float(F.cross_entropy(lg, lb) + M.lovasz_softmax(lg.softmax(1), lb)))


def test_ms_ssim_properties():
    torch.manual_seed(1)
    x = torch.rand(2, 3, 192, 208)
    assert float(M.ms_ssim(x, x)) == pytest.approx(1.0, abs=1e-6)
    y1, y2 = (x + 0.05 * torch.randn_like(x)).clamp(0, 1), (x + 0.3 * torch.randn_like(x)).clamp(0, 1)
    a, b = float(M.ms_ssim(x, y1)), float(M.ms_ssim(x, y2))
    assert 0.0 <= b < a < 1.0
    assert a == pytest.approx(float(M.ms_ssim(y1, x)), rel=1e-6)          # symmetric
    xx = x.clone().requires_grad_(True)
    (1 - M.ms_ssim(xx, y1)).backward()
    assert torch.isfinite(xx.grad).all() and float(xx.grad.abs().max()) > 0
    # single-scale sanity against a direct evaluation of the SSIM formula on constant patches:
    # identical constant images have l = cs = 1 at every scale
    c = torch.full((1, 3, 192, 192), 0.25)
    assert float(M.ms_ssim(c, c)) == pytest.approx(1.0, abs=1e-6)


def test_gaussian_focal_loss_against_loops():
    torch.manual_seed(2)
    p = torch.rand(2, 3, 5, 6).clamp(1e-4, 1 - 1e-4)
    t = torch.rand(2, 3, 5, 6)
    t[0, 1, 2, 3] = 1.0
    t[1, 0, 4, 1] = 1.0
    want = 0.0
    for pv, tv in zip(p.flatten().tolist(), t.flatten().tolist()):
        if tv == 1.0:
            want += -math.log(pv + 1e-12) * (1 - pv) ** 2
        want += -math.log(1 - pv + 1e-12) * pv ** 2 * (1 - tv) ** 4
    assert float(M.gaussian_focal_loss(p, t, avg_factor=2.0)) == pytest.approx(want / 2.0, rel=1e-5)


# ----------------------------------------------------------------------------- targets / decoding / NMS
def test_gaussian_radius_and_heatmap():
    # the three CornerNet cases for a 10 x 6 box at overlap 0.1 (closed forms)
    r = M.gaussian_radius((10.0, 6.0), 0.1)
    h, w, o = 10.0, 6.0, 0.1
    r1 = ((h + w) + math.sqrt((h + w) ** 2 - 4 * w * h * (1 - o) / (1 + o))) / 2
    r2 = (2 * (h + w) + math.sqrt(4 * (h + w) ** 2 - 16 * (1 - o) * w * h)) / 2
    r3 = (-2 * o * (h + w) + math.sqrt(4 * o * o * (h + w) ** 2 - 16 * o * (o - 1) * w * h)) / 2
    assert r == pytest.approx(min(r1, r2, r3))
    hm = torch.zeros(20, 30)
    M.draw_heatmap_gaussian(hm, (5, 7), 3)
    assert float(hm[7, 5]) == 1.0 and float(hm.max()) == 1.0
    assert float(hm[7, 8]) == pytest.approx(math.exp(-9 / (2 * (7 / 6) ** 2)), rel=1e-5)
    assert float(hm[7, 9]) == 0.0                                   # outside the 7 x 7 window
    M.draw_heatmap_gaussian(hm, (0, 0), 4)                          # clipped at the corner, merged by max
    assert float(hm[0, 0]) == 1.0 and float(hm[7, 5]) == 1.0
    M.draw_heatmap_gaussian(hm, (29, 19), 2)
    assert float(hm[19, 29]) == 1.0


def test_circle_nms():
    dets = np.array([[0.0, 0.0, 0.9], [1.0, 0.0, 0.8], [3.0, 0.0, 0.7], [3.5, 0.0, 0.95], [10.0, 10.0, 0.1]])
    # squared-distance threshold 1.5: (1,0) falls to (0,0); (3,0) falls to (3.5,0)
    assert M.circle_nms(dets, 1.5) == [3, 0, 4]
    assert M.circle_nms(dets, 0.1) == [3, 0, 1, 2, 4]
    assert M.circle_nms(dets, 1.5, post_max_size=2) == [3, 0]
    assert M.circle_nms(np.zeros((0, 3)), 1.0) == []
    # size-aware: two 4 x 2 boxes along x overlap in extent when 3 m apart (scale 1), not when 5 m apart
    sa = np.array([[0, 0, 4.0, 2.0, 0.0, 0.9], [3.0, 0, 4.0, 2.0, 0.0, 0.8], [5.0, 0.0, 4.0, 2.0, 0.0, 0.7],
                   [0.0, 2.5, 4.0, 2.0, 0.0, 0.6]])
    assert M.size_aware_circle_nms(sa, 1.0) == [0, 2, 3]
    rot = sa.copy()
    rot[:, 4] = math.pi / 2                                         # rotated: extents swap, x reach is 2 m
    assert M.size_aware_circle_nms(rot, 1.0) == [0, 1]             # (5, 0) is within 2 m of the kept (3, 0); (0, 2.5) within 4 m in y of (0, 0)


def _head(side=16, c0=8):
    bb, hd = M.reference_confs(dataclasses.replace(CFG_TINY), output_channels=c0, small_encoder=True)
    return M.BEVDepthHead(**hd), hd


def test_head_forward_targets_loss_and_decode():
    torch.manual_seed(0)
    head, hd = _head()
    side = hd["train_cfg"]["grid_size"][0] // 4
    x = torch.randn(2, 8, side, side)
    preds = head(x)
    assert len(preds) == 6 and preds[0][0]["heatmap"].shape == (2, 1, side, side)
    assert preds[1][0]["heatmap"].shape[1] == 2 and preds[0][0]["vel"].shape == (2, 2, side, side)
    assert float(preds[0][0]["heatmap"].mean()) == pytest.approx(-2.19, abs=0.8)          # init_bias
    batch = M.synthetic_batch(CFG_TINY, 2, seed=3, num_points=10, num_boxes=9)
    targets = head.get_targets(batch[4], batch[5])
    hm, anno, ind, mask = targets
    assert len(hm) == 6 and hm[0].shape == (2, 1, side, side) and anno[0].shape == (2, 500, 10)
    n_valid = sum(int(m.sum()) for m in mask)
    assert 0 < n_valid <= 18
    # every valid object's heatmap peak is exactly 1 at its cell, its regression target is the sub-cell offset
    for t in range(6):
        for b in range(2):
            for k in torch.nonzero(mask[t][b]).flatten().tolist():
                cell = int(ind[t][b, k])
                assert float(hm[t][b].reshape(hm[t].shape[1], -1)[:, cell].max()) == 1.0
                assert 0.0 <= float(anno[t][b, k, 0]) < 1.0 and 0.0 <= float(anno[t][b, k, 1]) < 1.0
                assert float(anno[t][b, k, 6] ** 2 + anno[t][b, k, 7] ** 2) == pytest.approx(1.0, abs=1e-5)
    loss = head.loss(targets, preds)
    assert torch.isfinite(loss) and float(loss) > 0
    loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in head.parameters())
    # decoding: plant one confident car at a known cell
    with torch.no_grad():
        preds = head(x)
        for pd in preds:
            pd[0]["heatmap"].fill_(-10.0)
        p0 = preds[0][0]
        p0["heatmap"][0, 0, 5, 9] = 6.0
        p0["reg"][0, :, 5, 9] = torch.tensor([0.25, 0.75])
        p0["rot"][0, :, 5, 9] = torch.tensor([1.0, 0.0])
        p0["dim"][0, :, 5, 9] = torch.tensor([0.5, 1.0, 0.2])
        p0["height"][0, :, 5, 9] = 0.3
        out = head.get_bboxes(preds)
    boxes, scores, labels = out[0]
    assert boxes.shape == (1, 9) and int(labels[0]) == 0 and float(scores[0]) == pytest.approx(torch.sigmoid(torch.tensor(6.0)).item())
    vs, rng = hd["bbox_coder"]["voxel_size"][0], hd["bbox_coder"]["pc_range"]
    assert float(boxes[0, 0]) == pytest.approx((9 + 0.25) * 4 * vs + rng[0], abs=1e-4)
    assert float(boxes[0, 1]) == pytest.approx((5 + 0.75) * 4 * vs + rng[1], abs=1e-4)
    assert float(boxes[0, 6]) == pytest.approx(math.pi / 2, abs=1e-5)
    assert boxes[0, 3:6].tolist() == pytest.approx([math.exp(0.5), math.exp(1.0), math.exp(0.2)], rel=1e-5)
    assert out[1][0].shape[0] == 0


def test_encoders_shapes_and_names():
    r50 = M.ResNet(depth=50)
    names = dict(r50.named_parameters())
    assert "layer1.0.conv3.weight" in names and "layer4.2.bn3.bias" in names and "conv1.weight" in names
    assert sum(p.numel() for p in r50.parameters()) == 23_508_032                       # torchvision resnet50 minus fc
    x = torch.randn(1, 3, 64, 96)
    feats = M.ResNet(depth=18)(x)
    assert [f.shape[1:] for f in feats] == [(64, 16, 24), (128, 8, 12), (256, 4, 6), (512, 2, 3)]
    neck = M.SECONDFPN(in_channels=[64, 128, 256, 512], upsample_strides=[0.5, 1, 2, 4], out_channels=[8, 8, 8, 8])
    assert neck(feats)[0].shape == (1, 32, 8, 12)                                     # stride 8


# ----------------------------------------------------------------------------- end to end (oracle operators)
class OracleOps:
    """Test-only stand-in for HotPath on CPU: every operator the module calls, computed by the oracle."""
    device = torch.device("cpu")

    def __init__(self, cfg):
        from vampire_amd.geometry import PathGeometry
        self.cfg, self.geo = cfg, PathGeometry(cfg)
        self.bounds = (cfg.x_bound_seg, cfg.y_bound_seg, cfg.z_bound_seg)

    def lift(self, depth, feat, lift_mats, use_depth=True):
        from oracle import aten_oracle as O
        if not use_depth:                               # base_bilinear.py:471-519: the D = 1 bilinear lift
            pix = O.ego_to_pixel(self.geo.voxel_coords, None, None, None, None, prepared=lift_mats)
            return O.lift_from_frustum_feats(feat.unsqueeze(3), pix, self.cfg.final_dim, self.cfg.d_bound, use_depth=False)
        return O.lift(depth, feat, self.geo.voxel_coords, None, None, None, None, self.cfg.final_dim, self.cfg.d_bound,
                      prepared=lift_mats)

    def render(self, dens, sem, base, rgb, beta, *, geom=None, render_mats=None):
        from oracle import aten_oracle as O
        c = self.cfg
        geom = torch.nan_to_num(O.frustum_to_ego(self.geo.frustum, None, None, None, None, prepared=render_mats), -1e3)
        return O.render(geom, dens, sem, base, rgb, seg_bounds=self.bounds, output_coords=self.geo.output_coords,
                        camera_mids=self.geo.camera_mids, bev_mids=self.geo.bev_mids, d_far=c.d_bound[1],
                        z_step_det=c.z_bound_det[2], num_classes=c.num_classes, density_mode=c.density_mode,
                        beta_param=beta, sdf_bias=c.sdf_bias, cat_seg=c.cat_seg)

    def sample_points(self, volume, points, *, padding="zeros", mask_outside=False, channel_last=False, **_):
        from oracle import aten_oracle as O
        out = O.sample_points(volume, points, self.bounds, padding, mask_outside)
        return out.transpose(1, 2) if channel_last else out

    def occupancy_queries(self, sem, dens, occ_coords, bda_mat, beta=None):
        from oracle import aten_oracle as O
        if bda_mat is None:                             # the static grid of the sibling backbones
            bda_mat = torch.eye(4).expand(sem.shape[0], 4, 4)
        return O.occupancy_queries(sem, dens, occ_coords, bda_mat, self.bounds, self.cfg.density_mode, beta, self.cfg.sdf_bias)


def _tiny_model(cfg):
    torch.manual_seed(0)
    bb, hd = M.reference_confs(cfg, output_channels=8, small_encoder=True)
    model = M.VAMPIRE2(bb, hd)
    model.backbone._hot = OracleOps(cfg)
    with torch.no_grad():
        model.backbone.density_conv.bias.fill_(cfg.sdf_bias)
    return model


def test_multitask_step_end_to_end_cpu():
    """configs[4] in miniature: R18 + SECONDFPN (stride-8 features beside the stride-4 frustum, as the
    reference's neck delivers) -> backbone -> BEV head, the collate_fn-shaped batch, all nine loss terms,
    one optimizer step."""
    cfg = dataclasses.replace(CFG_TINY, density_mode="sdf", final_dim=(192, 224), num_classes=6)      # (MS-SSIM: 5 scales of an 11-tap window need > 160 pixels)
    model = _tiny_model(cfg)
    batch = M.synthetic_batch(cfg, 2, seed=5, num_points=40, num_boxes=6)
    assert len(batch) == 20 and batch[0].shape == (2, 1, 6, 3, 192, 224) and batch[8].shape == (2, 1, 1, cfg.oY, cfg.oX)
    loss_fn = M.MultiTaskLoss(model, downsample_factor=4, upsample_factor=4, sdf_bias=cfg.sdf_bias)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4)
    before = [p.detach().clone() for p in model.parameters()]
    loss = M.multitask_step(model, loss_fn, batch, optimizer=opt, amp_dtype=None)
    assert torch.isfinite(loss)
    terms = loss_fn.last
    assert set(terms) >= {"detection", "depth", "seg", "rgb", "lidarseg", "sdf", "occ", "density"}
    for k, v in terms.items():
        assert torch.isfinite(torch.as_tensor(v)), k
        assert float(v) > 0, k
    # total = the reference's weighted sum (unit weights)
    tot = sum(float(terms[k]) for k in ("occ", "lidarseg", "detection", "depth", "seg", "rgb", "sdf", "density"))
    assert float(loss) == pytest.approx(tot, rel=1e-5)
    # every parameter got a finite gradient and the step moved the weights
    missing = [n for n, p in model.named_parameters() if p.grad is None]
    assert not missing, missing[:5]
    assert all(torch.isfinite(p.grad).all() for p in model.parameters())
    assert any(not torch.equal(a, b.detach()) for a, b in zip(before, model.parameters()))
    # eval-mode lidar-seg return (vampire2.py:62-63) and box decoding
    model.eval()
    with torch.no_grad():
        pts_logits, occ_logits, occ_density = model(batch[0], batch[1], inrange_pts=batch[11], lidar_seg=True)
        assert len(pts_logits) == 2 and occ_logits.shape == (2, 200, 200, 16, 6)
        out = model(batch[0], batch[1], inrange_pts=batch[11])
        dets = model.get_bboxes(out[0])
    assert len(dets) == 2 and dets[0][0].shape[1] == 9


# ----------------------------------------------------------------------------- data parallel (gloo, world size 2)
def _mt_worker(rank, world, port, out, mode):
    import dataclasses as dc
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), VAMP_GRAD_SYNC=mode)
    torch.set_num_threads(2)
    from vampire_amd import dist as vdist
    cfg = dc.replace(CFG_TINY, density_mode="sdf", final_dim=(192, 224), num_classes=6)
    vdist.init("gloo")
    model = _tiny_model(cfg)                                   # same seed -> same weights on both ranks
    ddp = vdist.wrap_ddp(model)
    assert isinstance(ddp, vdist.GradSync) == (mode == "hook")
    batch = M.synthetic_batch(cfg, 1, seed=vdist.shard_seed(0, rank), num_points=40, num_boxes=6)
    loss_fn = M.MultiTaskLoss(ddp, sdf_bias=cfg.sdf_bias)
    loss = M.multitask_step(ddp, loss_fn, batch, amp_dtype=None)
    g = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    out[rank] = (float(loss), g.numpy(), float(loss_fn.last["detection"]))
    vdist.shutdown()


@pytest.mark.parametrize("mode", ["hook", "ddp"])
def test_two_rank_multitask_step_gloo(mode):
    """configs[4] data-parallel on two gloo ranks (different shards), under the bucketed GradSync (the
    default) and under DistributedDataParallel: the ranks end the step with the same all-reduced
    gradients; the detection loss normalises by the cross-rank mean of positives
    (bev_depth_head.py:334-336 `reduce_mean`)."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = mp.Manager().dict()
    mp.spawn(_mt_worker, args=(2, port, out, mode), nprocs=2, join=True)
    (l0, g0, d0), (l1, g1, d1) = out[0], out[1]
    assert np.isfinite(l0) and np.isfinite(l1) and l0 != l1          # different shards, different losses
    assert np.isfinite(g0).all() and float(np.abs(g0).max()) > 0
    assert float(np.abs(g0 - g1).max()) <= 1e-6 * float(np.abs(g0).max())
