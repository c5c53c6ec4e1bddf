"""Pin the oracle (oracle/aten_oracle.py) against vectors produced by the reference itself
(tests/golden/make_golden.py ran /root/reference's BaseVAMPIRE2 / BaseBiLinear on CPU)."""
import dataclasses

import pytest
import torch

from conftest import load_golden, RENDER_VARIANTS, render_fixture_name
from oracle import aten_oracle as O
from vampire_amd.config import CFG_TINY
from vampire_amd.geometry import PathGeometry

GEO = PathGeometry(CFG_TINY)
SEG_BOUNDS = (CFG_TINY.x_bound_seg, CFG_TINY.y_bound_seg, CFG_TINY.z_bound_seg)


def test_buffers_shapes():
    c = CFG_TINY
    assert GEO.frustum.shape == (c.D, c.fH, c.fW, 4) == (21, 8, 22, 4)
    assert GEO.voxel_coords.shape == (c.vZ, c.vY, c.vX, 4) == (5, 16, 16, 4)
    # int() truncation quirk of the reference: (2.0 - -0.4) / 0.8 -> 2 cells
    assert GEO.output_coords.shape[0] == c.oZ == 2
    assert GEO.camera_mids.shape == (c.D - 1,)


def test_geometry_bitexact(tiny_common):
    """Bit-exact on the pinned prepared matrices; torch.inverse itself is CPU dependent in its
    last bits, so the recomputed-inverse path is only required to agree closely."""
    g = tiny_common
    geom = O.frustum_to_ego(GEO.frustum, None, None, None, None, prepared=g["render_mats"])
    assert torch.equal(geom, g["geom"])
    pix = O.ego_to_pixel(GEO.voxel_coords, None, None, None, None, prepared=g["lift_mats"])
    assert torch.equal(pix, g["pix"])
    geom2 = O.frustum_to_ego(GEO.frustum, g["sensor2ego"], g["intrin"], g["ida"], g["bda"])
    torch.testing.assert_close(geom2, g["geom"], rtol=1e-5, atol=1e-4)
    from vampire_amd.geometry import lift_matrices, render_matrices
    torch.testing.assert_close(lift_matrices(g["sensor2ego"], g["intrin"], g["ida"], g["bda"]),
                               g["lift_mats"], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(render_matrices(g["sensor2ego"], g["intrin"], g["ida"], g["bda"]),
                               g["render_mats"], rtol=1e-5, atol=1e-5)


def test_lift_forward_and_indices(tiny_common):
    g = tiny_common
    c = CFG_TINY
    vox = O.lift(g["depth"], g["feat"], GEO.voxel_coords, None, None, None, None,
                 c.final_dim, c.d_bound, prepared=g["lift_mats"])
    assert torch.equal(vox, g["lift"])
    valid, ix0, iy0, iz0 = O.lift_tap_indices(g["pix"], c.final_dim, c.d_bound, (c.D, c.fH, c.fW))
    assert torch.equal(valid.to(torch.uint8), g["lift_valid"])
    assert torch.equal(ix0.to(torch.int16), g["lift_ix0"])
    assert torch.equal(iy0.to(torch.int16), g["lift_iy0"])
    assert torch.equal(iz0.to(torch.int16), g["lift_iz0"])
    # the fixture exercises every mask path
    assert 0.05 < valid.float().mean() < 0.5
    assert (g["pix"][..., 2] < 0).any(), "fixture must contain points behind a camera"


def test_lift_backward(tiny_common):
    g = tiny_common
    c = CFG_TINY
    d = g["depth"].clone().requires_grad_(True)
    f = g["feat"].clone().requires_grad_(True)
    vox = O.lift(d, f, GEO.voxel_coords, None, None, None, None, c.final_dim, c.d_bound,
                 prepared=g["lift_mats"])
    vox.backward(g["g_lift"])
    torch.testing.assert_close(d.grad, g["grad_depth"], rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(f.grad, g["grad_feat"], rtol=1e-6, atol=1e-7)


def test_lift_bilinear_variant(tiny_common):
    """D == 1 lift (BaseBiLinear.get_voxel_feats, base_bilinear.py:471-519)."""
    g = load_golden("tiny_bilinear.npz")
    c = CFG_TINY
    pix = O.ego_to_pixel(GEO.voxel_coords, None, None, None, None, prepared=g["lift_mats"])
    f = g["feat"].clone().requires_grad_(True)
    vox = O.lift_from_frustum_feats(f.unsqueeze(3), pix, c.final_dim, c.d_bound, use_depth=False)
    assert torch.equal(vox, g["lift"])
    vox.backward(g["g_lift"])
    torch.testing.assert_close(f.grad, g["grad_feat"], rtol=1e-6, atol=1e-7)


NAMES = ["rgb_preds", "seg_logits_preds", "depth_preds", "bev_rgb_preds",
         "bev_seg_logits_preds", "bev_height_preds", "voxel_density", "voxel_output"]


@pytest.mark.parametrize("mode,cat_seg", RENDER_VARIANTS)
def test_render_forward_backward(tiny_common, mode, cat_seg):
    g = tiny_common
    r = load_golden(render_fixture_name(mode, cat_seg))
    c = dataclasses.replace(CFG_TINY, density_mode=mode, cat_seg=cat_seg)
    vols = [g[k].clone().requires_grad_(True)
            for k in ("density_feature", "semantic_logits", "base", "rgb")]
    beta = r["beta"].clone().reshape(()).requires_grad_(True) if mode == "sdf" else None
    geom = torch.nan_to_num(g["geom"], -1e3)
    outs = O.render(geom, *vols, seg_bounds=SEG_BOUNDS, output_coords=GEO.output_coords,
                    camera_mids=GEO.camera_mids, bev_mids=GEO.bev_mids, d_far=c.d_bound[1],
                    z_step_det=c.z_bound_det[2], num_classes=c.num_classes, density_mode=mode,
                    beta_param=beta, sdf_bias=c.sdf_bias, cat_seg=cat_seg)
    for name, o in zip(NAMES, outs):
        assert torch.equal(o, r[name]), name
    torch.autograd.backward(outs, [r["g_" + n] for n in NAMES])
    for k, v in zip(("density_feature", "semantic_logits", "base", "rgb"), vols):
        torch.testing.assert_close(v.grad, r["grad_" + k], rtol=1e-6, atol=1e-7)
    if mode == "sdf":
        torch.testing.assert_close(beta.grad.reshape(1), r["grad_beta"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("tag", ["smooth", "nonaffine"])
def test_render_smooth_and_nonaffine_fixtures(tag):
    """The oracle against the round-4 reference fixtures (smooth volumes; an `ida` that makes the frustum
    chain non-affine in the depth): forward bit for bit, gradients to 1e-6; the non-affine geometry also
    through the oracle's own frustum_to_ego on the prepared matrices."""
    r = load_golden(f"tiny_render_{tag}.npz")
    c = dataclasses.replace(CFG_TINY, density_mode="sdf", cat_seg=False)
    vols = [r[k].clone().requires_grad_(True) for k in ("density_feature", "semantic_logits", "base", "rgb")]
    beta = r["beta"].clone().reshape(()).requires_grad_(True)
    geom = torch.nan_to_num(O.frustum_to_ego(GEO.frustum, None, None, None, None, prepared=r["render_mats"]), -1e3)
    assert torch.equal(geom, r["geom"])
    outs = O.render(geom, *vols, seg_bounds=SEG_BOUNDS, output_coords=GEO.output_coords,
                    camera_mids=GEO.camera_mids, bev_mids=GEO.bev_mids, d_far=c.d_bound[1],
                    z_step_det=c.z_bound_det[2], num_classes=c.num_classes, density_mode="sdf",
                    beta_param=beta, sdf_bias=c.sdf_bias, cat_seg=False)
    for name, o in zip(NAMES, outs):
        assert torch.equal(o, r[name]), name
    torch.autograd.backward(outs, [r["g_" + n] for n in NAMES])
    for k, v in zip(("density_feature", "semantic_logits", "base", "rgb"), vols):
        torch.testing.assert_close(v.grad, r["grad_" + k], rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(beta.grad.reshape(1), r["grad_beta"], rtol=1e-5, atol=1e-6)


def test_render_indices(tiny_common):
    g = tiny_common
    c = CFG_TINY
    inside, ix0, iy0, iz0 = O.render_tap_indices(torch.nan_to_num(g["geom"], -1e3), SEG_BOUNDS,
                                                 (c.vZ, c.vY, c.vX))
    assert torch.equal(inside.to(torch.uint8), g["render_inside"])
    assert torch.equal(ix0.to(torch.int16), g["render_ix0"])
    assert torch.equal(iy0.to(torch.int16), g["render_iy0"])
    assert torch.equal(iz0.to(torch.int16), g["render_iz0"])
    assert 0.05 < inside.float().mean() < 0.9


def test_point_resampling_oracle():
    """SURVEY 8f N1 (bv2:576-609): the oracle's occupancy / lidar-point queries against the fixture
    replayed on the reference module's own occ_coords buffer, density module and bounds."""
    g = load_golden("tiny_points.npz")
    sem = g["semantic_logits"].clone().requires_grad_(True)
    dens = g["density_feature"].clone().requires_grad_(True)
    beta = g["beta"].clone().reshape(()).requires_grad_(True)
    occ_logits, occ_density = O.occupancy_queries(sem, dens, g["occ_sub"], g["bda"], SEG_BOUNDS, "sdf",
                                                  beta, CFG_TINY.sdf_bias)
    pts_logits = O.sample_points(sem, g["points"], SEG_BOUNDS, "border").permute(0, 2, 1)
    pts_sdf = O.sample_points(dens, g["points"], SEG_BOUNDS, "zeros", mask_outside=True)[:, 0]
    outs = dict(occ_logits=occ_logits, occ_density=occ_density, pts_logits=pts_logits, pts_sdf=pts_sdf)
    for k, v in outs.items():
        torch.testing.assert_close(v, g[k], rtol=1e-5, atol=1e-6, msg=k)
    torch.autograd.backward(list(outs.values()), [g["g_" + k] for k in outs])
    torch.testing.assert_close(sem.grad, g["grad_semantic_logits"], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(dens.grad, g["grad_density_feature"], rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(beta.grad.reshape(1), g["grad_beta"], rtol=1e-4, atol=1e-3)


def test_glue_oracle():
    """SURVEY 8f N2 (bv2:550, 627-630): depth softmax and density gate against the fixture replayed
    on the reference module's own mapping_along_depth layer."""
    g = load_golden("tiny_glue.npz")
    lg = g["depth_logits"].clone().requires_grad_(True)
    depth = O.depth_softmax(lg)
    torch.testing.assert_close(depth, g["depth"], rtol=1e-6, atol=1e-7)
    depth.backward(g["g_depth"])
    torch.testing.assert_close(lg.grad, g["grad_depth_logits"], rtol=1e-5, atol=1e-7)
    for mode in ("sdf", "naive"):
        vo = g[f"{mode}_voxel_output"].clone().requires_grad_(True)
        vd = g[f"{mode}_voxel_density"].clone().requires_grad_(True)
        out = O.density_gate(vo, vd, mode)
        torch.testing.assert_close(out, g[f"{mode}_gated"], rtol=1e-6, atol=1e-7)
        out.backward(g[f"{mode}_g_gated"])
        torch.testing.assert_close(vo.grad, g[f"{mode}_grad_voxel_output"], rtol=1e-6, atol=1e-7)
        torch.testing.assert_close(vd.grad, g[f"{mode}_grad_voxel_density"], rtol=1e-5, atol=1e-6)


def _hourglass_from_fixture(g, device="cpu"):
    from vampire_amd.backbone import Hourglass3D
    hg = Hourglass3D(4)
    hg.load_state_dict({k[2:]: v for k, v in g.items() if k.startswith("w_")})
    return hg.to(device)


def test_hourglass_and_resize_oracle():
    """SURVEY 8f N3, resize piece: the mirror of Hourglass3D (bv2:32-78) with the fixture's weights
    reproduces the reference class's outputs and gradients on CPU; the oracle's resize equals the
    recorded F.interpolate calls (bv2:66, 72)."""
    g = load_golden("tiny_hourglass.npz")
    hg = _hourglass_from_fixture(g)
    x = g["x"].clone().requires_grad_(True)
    out1, pre1, post1 = hg(x)
    out2, _, _ = hg(out1 + x, pre1, post1)
    for name, t in (("out1", out1), ("pre1", pre1), ("post1", post1), ("out2", out2)):
        torch.testing.assert_close(t, g[name], rtol=1e-5, atol=1e-6, msg=name)
    (out2 + x).backward(g["g_out"])
    torch.testing.assert_close(x.grad, g["grad_x"], rtol=1e-5, atol=1e-6)
    for k, v in hg.named_parameters():
        torch.testing.assert_close(v.grad, g["gw_" + k], rtol=1e-4, atol=1e-5, msg=k)
    for i in range(4):
        a = g[f"rs{i}_x"].clone().requires_grad_(True)
        r = O.upsample_trilinear(a, g[f"rs{i}_out"].shape[-3:])
        torch.testing.assert_close(r, g[f"rs{i}_out"], rtol=0, atol=0)
        r.backward(g[f"rs{i}_g"])
        torch.testing.assert_close(a.grad, g[f"rs{i}_grad"], rtol=1e-6, atol=1e-6)


def test_oracle_density_regimes_full_size():
    """The oracle's renderer at cfg-B against the REFERENCE's outputs in the regimes of
    tests/golden/regime_checksums.json (make_golden.py: make_regimes), forward only (the backward of
    one regime takes a minute on 8 cores): whole-tensor statistics of the eight outputs for the
    sigmoid density and for the reference's initial sdf regime."""
    import json
    import os
    from conftest import GOLDEN
    from vampire_amd.config import CFG_B
    from vampire_amd import synthetic
    with open(os.path.join(GOLDEN, "regime_checksums.json")) as f:
        ref = json.load(f)
    with open(os.path.join(GOLDEN, "full_checksums.json")) as f:
        rm = torch.tensor(json.load(f)["B"]["render_mats"], dtype=torch.float32)
    for regime in ("naive", "init"):
        cfg = dataclasses.replace(CFG_B, density_mode=ref[regime]["density_mode"])
        geo = PathGeometry(cfg)
        vols = list(synthetic.render_inputs(cfg, 1, seed=0))
        vols[0] = vols[0] * ref[regime]["density_scale"] + ref[regime]["density_shift"]
        with torch.no_grad():
            geom = torch.nan_to_num(O.frustum_to_ego(geo.frustum, None, None, None, None, rm), -1e3)
            outs = O.render(geom, *vols, seg_bounds=(cfg.x_bound_seg, cfg.y_bound_seg, cfg.z_bound_seg),
                            output_coords=geo.output_coords, camera_mids=geo.camera_mids, bev_mids=geo.bev_mids,
                            d_far=cfg.d_bound[1], z_step_det=cfg.z_bound_det[2], num_classes=cfg.num_classes,
                            density_mode=cfg.density_mode, beta_param=torch.tensor(ref[regime].get("beta", 0.1)),
                            sdf_bias=cfg.sdf_bias, cat_seg=False)
        for name, o in zip(NAMES, outs):
            st = ref[regime][name]
            assert list(o.shape) == st["shape"], (regime, name)
            tot = float(o.double().abs().sum())
            assert abs(tot - st["abs_sum"]) <= 1e-6 * st["abs_sum"] + 1e-6, (regime, name, tot, st["abs_sum"])
            assert abs(float(o.max()) - st["max"]) <= 1e-6 * max(1.0, abs(st["max"])), (regime, name)
