"""GPU parity: the HIP path (through the C ABI) against golden vectors produced by the
reference itself and against the oracle on identical seeded inputs.

Bars (BASELINE.json north_star): tap indices / masks bit-exact; values within 1e-4 fp32.
"""
import ctypes as C
import dataclasses
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, RENDER_VARIANTS, load_golden, render_fixture_name
from vampire_amd.config import CFG_A, CFG_B, CFG_D, CFG_TINY
from vampire_amd.geometry import PathGeometry, lift_matrices, render_matrices
from vampire_amd import synthetic
from vampire_amd import _capi

pytestmark = pytest.mark.gpu

ATOL = 1e-4   # north_star tolerance for rendered depth / semantics
NAMES = ["rgb_preds", "seg_logits_preds", "depth_preds", "bev_rgb_preds",
         "bev_seg_logits_preds", "bev_height_preds", "voxel_density", "voxel_output"]


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def hot(cfg, dev):
    from vampire_amd.ops import HotPath
    return HotPath(cfg, dev)


def close(a, b, atol=ATOL, rtol=1e-4, what="", scale=None, chan_dim=None):
    """|a - b| <= atol + rtol * |b|; with scale="max" the relative part uses max|b| (for
    gradients that are long fp32 sums of mixed-sign terms, where order of summation differs),
    taken per channel when chan_dim is given."""
    a, b = a.detach().cpu().float(), b.detach().cpu().float()
    if chan_dim is not None:
        for i in range(b.shape[chan_dim]):
            close(a.select(chan_dim, i), b.select(chan_dim, i), atol, rtol, f"{what}[ch {i}]", scale)
        return
    err = (a - b).abs()
    lim = atol + rtol * (b.abs().max() if scale == "max" else b.abs())
    assert bool((err <= lim).all()), f"{what}: max err {float(err.max()):.3e} (lim {atol}+{rtol}|x|)"


def tiny_mats(g, dev):
    """Prepared 4x4s pinned in the fixture (torch.inverse differs in its last bits between CPUs,
    so index parity is defined on identical prepared matrices)."""
    return g["lift_mats"].to(dev), (g["render_mats"].to(dev) if "render_mats" in g else None)


# --------------------------------------------------------------------------- lift
def test_lift_indices_bitexact_tiny(tiny_common, dev):
    g = tiny_common
    hp = hot(CFG_TINY, dev)
    lm, _ = tiny_mats(g, dev)
    valid, ix0, iy0, iz0 = hp.lift_indices(lm)
    assert torch.equal(valid.cpu(), g["lift_valid"])
    finite = torch.isfinite(g["pix"]).all(dim=-1)
    for got, key in ((ix0, "lift_ix0"), (iy0, "lift_iy0"), (iz0, "lift_iz0")):
        assert torch.equal(got.cpu()[finite], g[key][finite]), key


def _cull_covers_valid(hp, lm, valid, use_depth=True):
    """Every (voxel, camera) pair the chain finds valid has its camera's bit set in the cull word of the
    voxel's patch; returns the mean number of cameras a patch keeps."""
    words, (px, py) = hp.lift_cull_words(lm, use_depth=use_depth)
    B, N, Z, Y, X = valid.shape
    assert words.shape[1] == Z and words.shape[2] * py >= Y and words.shape[3] * px >= X
    per_voxel = words.repeat_interleave(py, dim=2).repeat_interleave(px, dim=3)[:, :, :Y, :X]     # [B, Z, Y, X]
    for n in range(N):
        missed = valid[:, n].bool() & ((per_voxel >> n) & 1).eq(0)
        assert not bool(missed.any()), f"camera {n}: {int(missed.sum())} valid voxels in culled patches"
    # bit 15: one inv(bda) per sample
    words = words[:, :, :(Y + py - 1) // py, :(X + px - 1) // px]          # (the launch grid's spare patches hold 0)
    same = (lm[:, :, 0] == lm[:, :1, 0]).flatten(1).all(dim=1)
    assert torch.equal(((words >> 15) & 1).flatten(1).amin(dim=1).bool(), same)
    assert torch.equal(((words >> 15) & 1).flatten(1).amax(dim=1).bool(), same)
    cams = sum(((words >> n) & 1) for n in range(N)).float()
    return float(cams.mean())


def test_lift_cull_words_tiny(tiny_common, dev):
    """The forward's conservative camera cull never drops a valid pair (the tiny fixture has a rotated,
    flipped and scaled bda and a camera partly behind the grid; tiny_bilinear is the D == 1 variant)."""
    g = tiny_common
    hp = hot(CFG_TINY, dev)
    lm, _ = tiny_mats(g, dev)
    _cull_covers_valid(hp, lm, g["lift_valid"].to(dev))
    gb = load_golden("tiny_bilinear.npz")
    lmb, _ = tiny_mats(gb, dev)
    validb = hp.lift_indices(lmb, use_depth=False)[0]
    _cull_covers_valid(hp, lmb, validb, use_depth=False)
    # a non-image-plane ida (depth leaks into u) switches the cull off for that camera
    lm2 = lm.clone()
    lm2[0, 1, 2, 0, 2] = 0.01
    w2, _ = hp.lift_cull_words(lm2)
    assert bool(((w2[0] >> 1) & 1).all())
    _cull_covers_valid(hp, lm2, hp.lift_indices(lm2)[0])


@pytest.mark.parametrize("name,cfg", [("A", CFG_A), ("B", CFG_B), ("D", CFG_D)])
def test_lift_cull_words_full_size(dev, name, cfg):
    """cfg-A / B / D with the synthetic rig (jittered, rotated bda, two samples): the cull keeps every valid
    pair and drops most cameras (about 1.5 of 6 per patch survive)."""
    hp = hot(cfg, dev)
    s2e, K, ida = synthetic.camera_rig(cfg, 2, jitter=2.0, seed=5)
    lm = lift_matrices(s2e, K, ida, synthetic.bda_matrix(2, rot_deg=7.0)).to(dev)
    valid = hp.lift_indices(lm)[0]
    kept = _cull_covers_valid(hp, lm, valid)
    assert kept < 2.2, kept
    s2e, K, ida = synthetic.camera_rig(cfg, 1)
    lm = lift_matrices(s2e, K, ida, synthetic.bda_matrix(1)).to(dev)
    kept = _cull_covers_valid(hp, lm, hp.lift_indices(lm)[0])
    assert kept < 2.2, kept


def test_lift_forward_tiny(tiny_common, dev):
    g = tiny_common
    hp = hot(CFG_TINY, dev)
    lm, _ = tiny_mats(g, dev)
    out = hp.lift(g["depth"].to(dev), g["feat"].to(dev), lm)
    close(out, g["lift"], atol=1e-5, what="lift")
    # the materialised-tensor entry point (get_voxel_feats signature)
    ff = (g["depth"].unsqueeze(2) * g["feat"].unsqueeze(3)).to(dev)
    out2 = hp.lift_dense(ff, lm)
    close(out2, g["lift"], atol=1e-5, what="lift_dense")


def test_lift_backward_tiny(tiny_common, dev):
    g = tiny_common
    hp = hot(CFG_TINY, dev)
    lm, _ = tiny_mats(g, dev)
    d = g["depth"].to(dev).requires_grad_(True)
    f = g["feat"].to(dev).requires_grad_(True)
    hp.lift(d, f, lm).backward(g["g_lift"].to(dev))
    # the fixture has voxels whose only samples are exact zeros: their 1/(0 + 1e-6) factor
    # (bv2:512) puts ~1e6-scaled mixed-sign terms into these sums
    close(d.grad, g["grad_depth"], atol=1e-5, rtol=1e-5, scale="max", what="grad_depth")
    close(f.grad, g["grad_feat"], atol=1e-5, rtol=1e-5, scale="max", chan_dim=2, what="grad_feat")
    # dense entry point: gradient w.r.t. the materialised tensor
    dd = g["depth"].to(dev).requires_grad_(True)
    fd = g["feat"].to(dev).requires_grad_(True)
    ff = dd.unsqueeze(2) * fd.unsqueeze(3)
    hp.lift_dense(ff, lm).backward(g["g_lift"].to(dev))
    close(dd.grad, g["grad_depth"], atol=1e-5, rtol=1e-5, scale="max", what="dense grad_depth")
    close(fd.grad, g["grad_feat"], atol=1e-5, rtol=1e-5, scale="max", chan_dim=2, what="dense grad_feat")


@pytest.mark.parametrize("C_", [8, 32, 48, 64])
def test_lift_channel_counts(tiny_common, dev, C_):
    """The lift at the channel counts the fixtures do not have (they are C = 4; cfg-A / B / D are 16): 8 (one chunk of
    8), 32 / 48 / 64 (chunks of 16; 48 takes the transpose tile's non-power-of-two path), forward with and without
    the pair emission and backward, against the oracle with autograd on the tiny geometry."""
    from oracle import aten_oracle as O
    g = tiny_common
    cfg = dataclasses.replace(CFG_TINY, mid_channels=C_)
    hp = hot(cfg, dev)
    geo = PathGeometry(cfg)
    lm, _ = tiny_mats(g, dev)
    gen = torch.Generator().manual_seed(100 + C_)
    B, N = g["depth"].shape[:2]
    feat = torch.randn(B, N, C_, cfg.fH, cfg.fW, generator=gen)
    feat[:, :, 1, ::3] = 0.0                     # exact zeros in a channel: the per-channel hit count (bv2:509-512)
    depth = g["depth"]
    dr, fr = depth.clone().requires_grad_(True), feat.clone().requires_grad_(True)
    ref = O.lift(dr, fr, geo.voxel_coords, None, None, None, None, cfg.final_dim, cfg.d_bound, prepared=g["lift_mats"])
    gout = torch.randn(ref.shape, generator=gen)
    # (voxels whose only samples are exact zeros carry a 1e6 factor, bv2:512: keep them out of the upstream gradient)
    with torch.no_grad():
        pix = O.ego_to_pixel(geo.voxel_coords, None, None, None, None, g["lift_mats"])
        valid, grid = O.lift_valid_and_grid(pix, cfg.final_dim, cfg.d_bound)
        ff = O.outer_depth_feat(depth, feat)
        sm = torch.nn.functional.grid_sample(ff.flatten(0, 1), grid.flatten(0, 1), align_corners=False).reshape(B, N, C_, *ref.shape[2:])
        fragile = ((sm.abs() < 1e-7) & valid.bool().unsqueeze(2)).any(dim=1)
    gout[fragile] = 0.0
    ref.backward(gout)
    with torch.no_grad():
        out0 = hp.lift(depth.to(dev), feat.to(dev), lm)
    close(out0, ref.detach(), atol=1e-5, what=f"lift C={C_} (no grad)")
    dd, fd = depth.to(dev).requires_grad_(True), feat.to(dev).requires_grad_(True)
    out = hp.lift(dd, fd, lm)
    close(out, ref.detach(), atol=1e-5, what=f"lift C={C_}")
    out.backward(gout.to(dev))
    close(dd.grad, dr.grad, atol=1e-5, rtol=1e-5, scale="max", what=f"grad_depth C={C_}")
    close(fd.grad, fr.grad, atol=1e-5, rtol=1e-5, scale="max", chan_dim=2, what=f"grad_feat C={C_}")


def test_lift_backward_rebuilds_overwritten_lists(tiny_common, dev):
    """Two forwards on one workspace, then the backward of the FIRST: the pairs the second forward left are not the
    first one's, so that backward builds its own cell lists (lift_pairs_kernel: projection + depth samples) -- same
    gradients as an undisturbed forward + backward."""
    g = tiny_common
    hp = hot(CFG_TINY, dev)
    lm, _ = tiny_mats(g, dev)
    d1, f1 = g["depth"].to(dev).requires_grad_(True), g["feat"].to(dev).requires_grad_(True)
    out1 = hp.lift(d1, f1, lm)
    gen = torch.Generator(device=dev).manual_seed(9)
    d2 = torch.rand(d1.shape, device=dev, generator=gen).requires_grad_(True)
    f2 = torch.randn(f1.shape, device=dev, generator=gen).requires_grad_(True)
    out2 = hp.lift(d2, f2, lm.flip(1))                     # other values, other camera order: other pairs
    out1.backward(g["g_lift"].to(dev))
    close(d1.grad, g["grad_depth"], atol=1e-5, rtol=1e-5, scale="max", what="grad_depth behind a second forward")
    close(f1.grad, g["grad_feat"], atol=1e-5, rtol=1e-5, scale="max", chan_dim=2, what="grad_feat behind a second forward")
    out2.sum().backward()                                  # (its lists are gone too by now)
    assert bool(torch.isfinite(d2.grad).all()) and bool(torch.isfinite(f2.grad).all())


def test_lift_bilinear_variant(dev):
    g = load_golden("tiny_bilinear.npz")
    hp = hot(CFG_TINY, dev)
    lm, _ = tiny_mats(g, dev)
    f = g["feat"].to(dev).requires_grad_(True)
    out = hp.lift(None, f, lm, use_depth=False)
    close(out, g["lift"], atol=1e-5, what="bilinear lift")
    out.backward(g["g_lift"].to(dev))
    close(f.grad, g["grad_feat"], atol=1e-5, rtol=1e-5, scale="max", chan_dim=2, what="bilinear grad_feat")


def test_lift_bf16_inputs(tiny_common, dev):
    """bf16 depth/feat are promoted to fp32 in-kernel: result must equal
    the fp32 path run on the bf16-rounded values."""
    g = tiny_common
    hp = hot(CFG_TINY, dev)
    lm, _ = tiny_mats(g, dev)
    d16, f16 = g["depth"].to(dev).bfloat16(), g["feat"].to(dev).bfloat16()
    a = hp.lift(d16, f16, lm)
    b = hp.lift(d16.float(), f16.float(), lm)
    assert torch.equal(a, b)


def channel_last(feat):
    """The same [B, N, C, fH, fW] values in [B, N, fH, fW, C] memory (a torch.channels_last producer's output)."""
    return feat.permute(0, 1, 3, 4, 2).contiguous().permute(0, 1, 4, 2, 3)


@pytest.mark.parametrize("C_", [4, 16, 48])
def test_lift_channel_last_features_tiny(tiny_common, dev, C_):
    """VAMP_LIFTFWD_FEAT_CHANNEL_LAST (round 6): features handed over in channel-last memory go in zero-copy -- no
    transposing first launch, the forward's workgroups form their camera cull words themselves -- and every result has
    the bits of the [B, N, C, fH, fW] call: forward with and without the pair emission, the cell-list backward (the
    gradient comes back in the features' own layout), the logits entry, the D = 1 variant; the float-atomic splat to
    rounding."""
    from vampire_amd.ops import _is_channel_last
    g = tiny_common
    cfg = dataclasses.replace(CFG_TINY, mid_channels=C_)
    hp = hot(cfg, dev)
    lm, _ = tiny_mats(g, dev)
    gen = torch.Generator().manual_seed(500 + C_)
    B, N = g["depth"].shape[:2]
    feat = (g["feat"] if C_ == 4 else torch.randn(B, N, C_, cfg.fH, cfg.fW, generator=gen)).to(dev)
    depth = g["depth"].to(dev)
    fcl = channel_last(feat)
    assert _is_channel_last(fcl) and not _is_channel_last(feat) and torch.equal(fcl, feat)
    with torch.no_grad():
        assert torch.equal(hp.lift(depth, fcl, lm), hp.lift(depth, feat, lm)), "no-grad forward"
        assert torch.equal(hp.lift(None, fcl, lm, use_depth=False), hp.lift(None, feat, lm, use_depth=False)), "D = 1 forward"
    gout = torch.randn(B, C_, cfg.vZ, cfg.vY, cfg.vX, generator=gen).to(dev)
    res = {}
    for tag, f0 in (("cl", fcl), ("cf", feat)):
        for impl in ("cell", "v1"):
            hp.impl["lift_bwd"] = impl
            d, f = depth.clone().requires_grad_(True), f0.clone(memory_format=torch.preserve_format).requires_grad_(True)
            out = hp.lift(d, f, lm)
            out.backward(gout)
            res[tag, impl] = (out.detach(), d.grad, f.grad)
        hp.impl["lift_bwd"] = "cell"
        lg = depth.clamp_min(1e-30).log().requires_grad_(True)
        f = f0.clone(memory_format=torch.preserve_format).requires_grad_(True)
        out = hp.lift_logits(lg, f, lm)
        out.backward(gout)
        res[tag, "logits"] = (out.detach(), lg.grad, f.grad)
    assert _is_channel_last(res["cl", "cell"][2]), "grad_feat comes back channel-last"
    # (forward: same bits; the gradients are fp32 sums over a cell's pairs in the order the fill's atomics handed out
    # their slots, which differs from run to run in the last bits whatever the layout)
    for impl in ("cell", "logits"):
        assert torch.equal(res["cl", impl][0], res["cf", impl][0]), f"{impl}: forward differs between the feature layouts"
        for nm, a, b_ in zip(("grad_depth", "grad_feat"), res["cl", impl][1:], res["cf", impl][1:]):
            close(a, b_, atol=1e-7, rtol=2e-6, scale="max", what=f"{impl}: {nm} between the feature layouts")
    for nm, a, b_ in zip(("out", "grad_depth", "grad_feat"), res["cl", "v1"], res["cf", "cell"]):
        close(a, b_, atol=1e-6, rtol=1e-5, scale="max", what=f"splat on channel-last features, {nm}")


def test_lift_channel_last_features_full_size(dev):
    """cfg-B, two samples with jittered rigs and a rotated bda: the inline cull words are the first launch's (same
    outputs bit for bit, i.e. no valid pair lost), forward and backward."""
    cfg = CFG_B
    hp = hot(cfg, dev)
    s2e, K, ida = synthetic.camera_rig(cfg, 2, jitter=2.0, seed=5)
    lm = lift_matrices(s2e, K, ida, synthetic.bda_matrix(2, rot_deg=4.0, scale=0.98)).to(dev)
    depth, feat = synthetic.lift_inputs(cfg, 2, seed=9, device=dev)
    fcl = channel_last(feat)
    with torch.no_grad():
        assert torch.equal(hp.lift(depth, fcl, lm), hp.lift(depth, feat, lm)), "no-grad forward (cooperative kernel)"
    gen = torch.Generator(device=dev).manual_seed(77)
    gout = torch.randn(2, cfg.mid_channels, cfg.vZ, cfg.vY, cfg.vX, device=dev, generator=gen)
    got = []
    for f0 in (fcl, feat):
        d, f = depth.clone().requires_grad_(True), f0.clone(memory_format=torch.preserve_format).requires_grad_(True)
        out = hp.lift(d, f, lm)
        out.backward(gout)
        got.append((out.detach(), d.grad, f.grad))
    assert torch.equal(got[0][0], got[1][0]), "training forward differs between the feature layouts"
    for nm, a, b_ in zip(("grad_depth", "grad_feat"), got[0][1:], got[1][1:]):
        close(a, b_, atol=1e-7, rtol=2e-6, scale="max", what=f"{nm} between the feature layouts")


@pytest.mark.parametrize("ldtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_lift_logits_tiny(tiny_common, dev, ldtype):
    """N2 producer fusion (bv2:550 + 553): the lift fed with depth LOGITS -- softmax inside the operand
    launch, softmax backward inside the gather -- against torch.softmax -> lift through autograd, and
    against the golden lift at the logits whose softmax is the fixture's depth."""
    g = tiny_common
    hp = hot(CFG_TINY, dev)
    lm, _ = tiny_mats(g, dev)
    lg0 = g["depth"].to(dev).clamp_min(1e-30).log()              # softmax(log p) = p for a normalised p
    if ldtype == torch.float32:
        close(hp.lift_logits(lg0, g["feat"].to(dev), lm), g["lift"], atol=1e-5, what="lift_logits vs golden")
    gen = torch.Generator(device=dev).manual_seed(3)
    lg = (torch.randn(lg0.shape, device=dev, generator=gen) * 2).to(ldtype)
    a_l = lg.clone().requires_grad_(True); a_f = g["feat"].to(dev).requires_grad_(True)
    b_l = lg.clone().requires_grad_(True); b_f = g["feat"].to(dev).requires_grad_(True)
    out_a = hp.lift_logits(a_l, a_f, lm)
    out_b = hp.lift(b_l.float().softmax(dim=2), b_f, lm)
    close(out_a, out_b, atol=1e-6, rtol=1e-5, what="lift_logits forward")
    out_a.backward(g["g_lift"].to(dev)); out_b.backward(g["g_lift"].to(dev))
    assert a_l.grad.dtype == ldtype
    tol = dict(atol=1e-5, rtol=1e-5) if ldtype == torch.float32 else dict(atol=1e-5, rtol=1e-2)
    close(a_l.grad, b_l.grad, scale="max", what="grad logits", **tol)
    close(a_f.grad, b_f.grad, atol=1e-5, rtol=1e-5, scale="max", chan_dim=2, what="grad feat")
    # the atomic-splat cross-check implementation behind the same entry
    hp.impl["lift_bwd"] = "v1"
    c_l = lg.clone().requires_grad_(True); c_f = g["feat"].to(dev).requires_grad_(True)
    hp.lift_logits(c_l, c_f, lm).backward(g["g_lift"].to(dev))
    close(c_l.grad, b_l.grad, scale="max", what="v1 grad logits", **tol)
    # no-grad call: no hit words, same values
    with torch.no_grad():
        assert torch.equal(hp.lift_logits(lg, g["feat"].to(dev), lm), out_a)


@pytest.mark.parametrize("wpp", [0, 4], ids=["wpp1", "wpp4"])
def test_lift_logits_full_size(dev, wpp):
    """Same at cfg-B, B=2 (jittered rigs + bda), halves on two streams as in the training step."""
    cfg = CFG_B
    hp = hot(cfg, dev)
    hp.impl["lift_wpp"] = wpp
    s2e, K, ida = synthetic.camera_rig(cfg, 2, jitter=2.0, seed=11)
    lm = lift_matrices(s2e, K, ida, synthetic.bda_matrix(2, rot_deg=-6.0, scale=1.02)).to(dev)
    gen = torch.Generator(device=dev).manual_seed(8)
    _, feat = synthetic.lift_inputs(cfg, 2, seed=6, device=dev)
    lg = torch.randn(2, feat.shape[1], cfg.D, cfg.fH, cfg.fW, device=dev, generator=gen) * 3
    a_l = lg.clone().requires_grad_(True); a_f = feat.clone().requires_grad_(True)
    b_l = lg.clone().requires_grad_(True); b_f = feat.clone().requires_grad_(True)
    out_a = hp.lift_logits(a_l, a_f, lm)
    out_b = hp.lift(b_l.softmax(dim=2), b_f, lm)
    close(out_a, out_b, atol=1e-6, rtol=1e-5, what="forward")
    go = torch.randn(out_a.shape, device=dev, generator=gen)
    out_a.backward(go); out_b.backward(go)
    assert float(b_l.grad.abs().max()) > 0
    close(a_l.grad, b_l.grad, atol=1e-7, rtol=2e-5, scale="max", what="grad logits")
    close(a_f.grad, b_f.grad, atol=1e-6, rtol=2e-5, scale="max", chan_dim=2, what="grad feat")
    # softmax rows: the gradient of the logits sums to zero over D
    assert float(a_l.grad.sum(2).abs().max()) <= 1e-4 * float(a_l.grad.abs().max()) * cfg.D ** 0.5 + 1e-7


# --------------------------------------------------------------------------- render
def test_frustum_geometry_bitexact(tiny_common, dev):
    g = tiny_common
    hp = hot(CFG_TINY, dev)
    _, rm = tiny_mats(g, dev)
    geom = hp.frustum_geometry(rm)
    assert torch.equal(geom.cpu(), torch.nan_to_num(g["geom"], -1e3))


def test_render_indices_bitexact_tiny(tiny_common, dev):
    g = tiny_common
    hp = hot(CFG_TINY, dev)
    _, rm = tiny_mats(g, dev)
    for kw in (dict(render_mats=rm), dict(geom=torch.nan_to_num(g["geom"], -1e3).to(dev))):
        inside, ix0, iy0, iz0 = hp.render_indices(**kw)
        assert torch.equal(inside.cpu(), g["render_inside"])
        m = g["render_inside"].bool()
        for got, key in ((ix0, "render_ix0"), (iy0, "render_iy0"), (iz0, "render_iz0")):
            assert torch.equal(got.cpu()[m], g[key][m]), key


@pytest.mark.parametrize("mode,cat_seg", RENDER_VARIANTS)
@pytest.mark.parametrize("use_geom", [False, True])
def test_render_forward_backward_tiny(tiny_common, dev, mode, cat_seg, use_geom):
    g = tiny_common
    r = load_golden(render_fixture_name(mode, cat_seg))
    cfg = dataclasses.replace(CFG_TINY, density_mode=mode, cat_seg=cat_seg)
    hp = hot(cfg, dev)
    _, rm = tiny_mats(g, dev)
    vols = [g[k].to(dev).requires_grad_(True)
            for k in ("density_feature", "semantic_logits", "base", "rgb")]
    beta = (r["beta"].reshape(()).to(dev).requires_grad_(True) if mode == "sdf" else None)
    kw = (dict(geom=torch.nan_to_num(g["geom"], -1e3).to(dev)) if use_geom
          else dict(render_mats=rm))
    outs = hp.render(*vols, beta, **kw)
    for name, o in zip(NAMES, outs):
        close(o, r[name], what=name)
    torch.autograd.backward(outs, [r["g_" + n].to(dev) for n in NAMES])
    for k, v in zip(("density_feature", "semantic_logits", "base", "rgb"), vols):
        close(v.grad, r["grad_" + k], atol=1e-5, rtol=1e-5, scale="max", what="grad_" + k)
    if mode == "sdf":
        close(beta.grad.reshape(1), r["grad_beta"], atol=1e-3, rtol=1e-3, what="grad_beta")


def test_render_non_finite_volume_entries(tiny_common, dev, monkeypatch):
    """nan / +-inf voxels: the sampled features go through nan_to_num (bv2:421) before density and
    compositing.  The kernels sanitise only where a sample is not finite (fast path otherwise);
    forward against the oracle, camera-branch backward against the v1 implementation."""
    from oracle import aten_oracle as O
    g = tiny_common
    cfg = dataclasses.replace(CFG_TINY, density_mode="sdf", cat_seg=False)
    hp = hot(cfg, dev)
    geo = PathGeometry(cfg)
    _, rm = tiny_mats(g, dev)
    vols = [g[k].clone() for k in ("density_feature", "semantic_logits", "base", "rgb")]
    gen = torch.Generator().manual_seed(4)
    # +-inf only in the density feature (it saturates the density); in the composited channels an
    # inf becomes +-FLT_MAX, whose sums are not comparable at any tolerance
    for t, vals in zip((vols[0], vols[0], vols[1], vols[3]),
                       ((float("nan"), float("inf")), (float("nan"), -float("inf")),
                        (float("nan"), float("nan")), (float("nan"), float("nan")))):
        flat = t.view(-1)
        idx = torch.randperm(flat.numel(), generator=gen)[:24]
        flat[idx[:12]] = vals[0]
        flat[idx[12:]] = vals[1]
    beta = torch.tensor(0.1)
    geom = torch.nan_to_num(g["geom"], -1e3)
    ref = O.render(geom, *vols, seg_bounds=(cfg.x_bound_seg, cfg.y_bound_seg, cfg.z_bound_seg),
                   output_coords=geo.output_coords, camera_mids=geo.camera_mids, bev_mids=geo.bev_mids,
                   d_far=cfg.d_bound[1], z_step_det=cfg.z_bound_det[2], num_classes=cfg.num_classes,
                   density_mode="sdf", beta_param=beta, sdf_bias=cfg.sdf_bias, cat_seg=False)

    def run(impl):
        hp.impl["cam_bwd"] = impl
        dv = [v.to(dev).requires_grad_(True) for v in vols]
        b = beta.to(dev).requires_grad_(True)
        outs = hp.render(*dv, b, render_mats=rm)
        gg = torch.Generator(device=dev).manual_seed(3)
        gs = [torch.randn(o.shape, device=dev, generator=gg) for o in outs]
        for i in (3, 4, 5, 6, 7):
            gs[i].zero_()
        torch.autograd.backward(outs, gs)
        return outs, [v.grad for v in dv], b.grad

    outs, grads, _ = run("cell")
    for name, o, r in list(zip(NAMES, outs, ref))[:3]:          # camera branch
        close(o, r, atol=2e-4, rtol=1e-4, what="non-finite volume: " + name)
    _, grads1, _ = run("v1")
    # the BEV branch has no nan_to_num in the reference (bv2:442-461), so its backward puts nan into
    # the gradients around the non-finite voxels in both runs: same places, and equal elsewhere
    for name, a, b in zip(("density_feature", "semantic_logits", "base", "rgb"), grads, grads1):
        fin = torch.isfinite(b)
        assert torch.equal(torch.isfinite(a), fin), name
        assert float(fin.float().mean()) > 0.5, name
        close(torch.where(fin, a, 0.0), torch.where(fin, b, 0.0), atol=1e-5, rtol=2e-5, scale="max",
              what="non-finite volume: cell vs v1 grad_" + name)


# --------------------------------------------------------------------------- producer / consumer glue
def test_glue_tiny(dev):
    """depth softmax (bv2:550) and density gate (bv2:627-630) against the fixture made
    with the reference module's own layers: values and input gradients."""
    g = load_golden("tiny_glue.npz")
    hp = hot(CFG_TINY, dev)
    lg = g["depth_logits"].to(dev).requires_grad_(True)
    depth = hp.depth_softmax(lg)
    close(depth, g["depth"], atol=1e-6, rtol=1e-5, what="depth softmax")
    close(depth.sum(1), torch.ones_like(depth[:, 0]), atol=1e-6, rtol=0, what="softmax sums to one")
    depth.backward(g["g_depth"].to(dev))
    close(lg.grad, g["grad_depth_logits"], atol=1e-6, rtol=1e-5, what="grad_depth_logits")
    for mode in ("sdf", "naive"):
        hpm = hot(dataclasses.replace(CFG_TINY, density_mode=mode), dev)
        vo = g[f"{mode}_voxel_output"].to(dev).requires_grad_(True)
        vd = g[f"{mode}_voxel_density"].to(dev).requires_grad_(True)
        out = hpm.density_gate(vo, vd)
        close(out, g[f"{mode}_gated"], atol=1e-6, rtol=1e-5, what=mode + " gated")
        out.backward(g[f"{mode}_g_gated"].to(dev))
        close(vo.grad, g[f"{mode}_grad_voxel_output"], atol=1e-6, rtol=1e-5, what=mode + " grad_voxel_output")
        close(vd.grad, g[f"{mode}_grad_voxel_density"], atol=1e-5, rtol=1e-5, what=mode + " grad_voxel_density")


@pytest.mark.parametrize("shape,dtype", [((6, 86, 64, 176), torch.float32), ((6, 86, 64, 176), torch.bfloat16),
                                         ((3, 5, 7, 9), torch.float32), ((2, 1, 1, 70), torch.float32),
                                         ((1, 131, 3, 33), torch.float32)])
def test_depth_softmax_shapes(dev, shape, dtype):
    """Full-size cfg-B depth head and ragged shapes (pixel count not a multiple of the 64-pixel
    tile, fewer depth bins than waves) against the oracle, forward and backward; -inf logits."""
    from oracle import aten_oracle as O
    gen = torch.Generator().manual_seed(3)
    lg = (torch.randn(shape, generator=gen) * 4.0).to(dtype)
    if shape[1] > 2:
        lg[0, 1, 0, 0] = float("-inf")
    up = torch.randn(shape, generator=gen)
    ref_in = lg.float().clone().requires_grad_(True)
    ref = O.depth_softmax(ref_in)
    ref.backward(up)
    x = lg.to(dev).requires_grad_(True)
    got = hot(CFG_B, dev).depth_softmax(x)
    assert got.dtype == torch.float32 and x.grad is None
    close(got, ref, atol=1e-6, rtol=1e-5, what="softmax")
    got.backward(up.to(dev))
    assert x.grad.dtype == dtype
    close(x.grad, ref_in.grad, atol=1e-6 if dtype == torch.float32 else 1e-3,
          rtol=1e-5 if dtype == torch.float32 else 1e-2, scale="max", what="softmax grad")


def test_density_gate_full_size(dev):
    """cfg-B BEV block [1, 16, 10, 200, 200] (and cat_seg's 34 channels, batch 2) against the oracle."""
    from oracle import aten_oracle as O
    gen = torch.Generator().manual_seed(4)
    for mode, shape in (("sdf", (1, 16, 10, 200, 200)), ("naive", (2, 34, 10, 200, 200))):
        vo = torch.randn(shape, generator=gen)
        vd = torch.rand((shape[0], 1) + shape[2:], generator=gen) * 4.0
        up = torch.randn(shape, generator=gen)
        a, b = vo.clone().requires_grad_(True), vd.clone().requires_grad_(True)
        O.density_gate(a, b, mode).backward(up)
        x, y = vo.to(dev).requires_grad_(True), vd.to(dev).requires_grad_(True)
        out = hot(dataclasses.replace(CFG_B, density_mode=mode), dev).density_gate(x, y)
        close(out, O.density_gate(vo, vd, mode), atol=1e-6, rtol=1e-5, what="gated")
        out.backward(up.to(dev))
        close(x.grad, a.grad, atol=1e-6, rtol=1e-5, what="grad_voxel_output")
        close(y.grad, b.grad, atol=1e-5, rtol=1e-5, scale="max", what="grad_voxel_density")


# --------------------------------------------------------------------------- HIP graph capture
@pytest.mark.parametrize("mode", ["sdf", "naive"])
@pytest.mark.parametrize("shape", [(2, 16, 10, 37, 41, 80), (1, 9, 6, 16, 16, 8), (1, 16, 10, 256, 256, 80)],
                         ids=["ragged", "tiny", "ref-grid"])
def test_gate_conv1x1(dev, mode, shape):
    import torch.nn.functional as F
    """N2 consumer fusion (bv2:627-632): gate + `voxel_output` 1x1 conv in one matrix-core kernel each way
    against the reference's aten expression, all five gradients; ragged sizes exercise the cell / channel
    padding, `ref-grid` is the reference's 256 x 256 x 10 det grid with 16 -> 80 channels."""
    import dataclasses
    B, C_, oZ, oY, oX, cout = shape
    hp = hot(dataclasses.replace(CFG_TINY, density_mode=mode), dev)
    assert hp.gate_conv1x1_supported(C_, oZ, cout)
    gen = torch.Generator(device=dev).manual_seed(5)
    rnd = lambda *s_: torch.randn(*s_, device=dev, generator=gen)
    vo, vd = rnd(B, C_, oZ, oY, oX), rnd(B, 1, oZ, oY, oX)
    w, bs = rnd(cout, C_ * oZ, 1, 1) * 0.1, rnd(cout)
    go = rnd(B, cout, oY, oX)
    leaves_a = [t.clone().requires_grad_(True) for t in (vo, vd, w, bs)]
    leaves_b = [t.clone().double().requires_grad_(True) for t in (vo, vd, w, bs)]
    out = hp.gate_conv1x1(*leaves_a)
    gate = leaves_b[1].tanh() if mode == "sdf" else leaves_b[1]
    ref = F.conv2d((leaves_b[0] * gate).reshape(B, C_ * oZ, oY, oX), leaves_b[2], leaves_b[3])
    close(out, ref, atol=1e-5, rtol=1e-5, scale="max", what="gate_conv forward")
    out.backward(go); ref.backward(go.double())
    for a, b_, name in zip(leaves_a, leaves_b, ("voxel_output", "voxel_density", "weight", "bias")):
        close(a.grad, b_.grad, atol=1e-6, rtol=2e-5, scale="max", what="grad " + name)
    # deterministic: same bits on a second run
    leaves_c = [t.clone().requires_grad_(True) for t in (vo, vd, w, bs)]
    hp.gate_conv1x1(*leaves_c).backward(go)
    assert all(torch.equal(a.grad, c.grad) for a, c in zip(leaves_a, leaves_c))
    # no bias
    o2 = hp.gate_conv1x1(vo, vd, w, None)
    close(o2, ref - leaves_b[3].view(1, -1, 1, 1), atol=1e-5, rtol=1e-5, scale="max", what="no bias")
    assert not hp.gate_conv1x1_supported(34, 10, 80)      # cat_seg at the reference sizes: aten path


@pytest.mark.parametrize("cfg,batch", [(CFG_TINY, 2), (CFG_B, 1)], ids=["tiny", "cfg-B"])
def test_step_is_graph_capturable(dev, cfg, batch):
    """The whole step (both streams, the prepare passes, workspaces, memsets) captures into one
    HIP graph; replays reproduce the eager gradients (up to the order in which a cell's records are
    summed, which the integer rank atomics leave open: 1e-6 of the largest gradient)."""
    from vampire_amd.step import LiftRenderStep, SyntheticBatch, train_step
    model = LiftRenderStep(cfg, dev)
    data = SyntheticBatch(cfg, batch, dev)

    def step():
        model.zero_grad(set_to_none=True)
        train_step(model, data)

    def grads():
        return [data.depth.grad, data.feat.grad] + [v.grad for v in data.vols] + [model.beta.grad]

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    ref = [g.detach().clone() for g in grads()]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        step()
    for _ in range(3):
        for g in grads():
            g.zero_()
        graph.replay()
        torch.cuda.synchronize()
        for name, a, b in zip(("depth", "feat", "density_feature", "semantic_logits", "base", "rgb", "beta"),
                              grads(), ref):
            err, top = float((a - b).abs().max()), float(b.abs().max())
            assert err <= (1e-5 if name == "beta" else 1e-6) * top, "graph replay differs from eager: grad_%s, max |diff| %.3e of %.3e" % (
                name, err, top)


@pytest.mark.parametrize("cfg,batch", [(CFG_TINY, 2), (CFG_B, 1)], ids=["tiny", "cfg-B"])
def test_deferred_lift_scan_matches_the_scan_in_the_forward(dev, cfg, batch):
    """A training lift forward leaves the scan of its pair cells to the render forward's prepare step, which scans both
    cell lists in one launch (vamp_render_camera_prepare_with_lift): same step gradients as with the scan inside the
    lift's own call -- up to the order of a cell's records (1e-6 of the largest gradient)."""
    from vampire_amd.step import LiftRenderStep, SyntheticBatch, train_step

    def grads_of(defer):
        model = LiftRenderStep(cfg, dev)
        model.hp.impl["defer_lift_scan"] = defer
        data = SyntheticBatch(cfg, batch, dev)
        for _ in range(2):                                  # (the second step runs on workspaces the first has used)
            model.zero_grad(set_to_none=True)
            train_step(model, data)
        assert model.hp._lift_scan_pending is None          # somebody has run it
        torch.cuda.synchronize()
        return [data.depth.grad, data.feat.grad] + [v.grad for v in data.vols] + [model.beta.grad]

    for name, a, b in zip(("depth", "feat", "density_feature", "semantic_logits", "base", "rgb", "beta"),
                          grads_of(True), grads_of(False)):
        err, top = float((a - b).abs().max()), float(b.abs().max())
        assert err <= (1e-5 if name == "beta" else 1e-6) * top, "grad_%s: max |diff| %.3e of %.3e" % (name, err, top)


# --------------------------------------------------------------------------- trilinear resize (UNet)
def test_resize_and_hourglass_tiny(dev):
    """SURVEY 8f N3, resize piece (bv2:66, 72): the HIP resize against the recorded F.interpolate
    calls (up, ragged, down, depth-1 source), forward and backward; then the Hourglass3D mirror on
    the GPU (HIP resize inside) against the reference class's outputs and gradients."""
    from vampire_amd.ops import upsample_trilinear
    from test_oracle_golden import _hourglass_from_fixture
    g = load_golden("tiny_hourglass.npz")
    for i in range(4):
        a = g[f"rs{i}_x"].to(dev).requires_grad_(True)
        r = upsample_trilinear(a, g[f"rs{i}_out"].shape[-3:])
        close(r, g[f"rs{i}_out"], atol=1e-6, rtol=1e-6, what=f"resize {i}")
        r.backward(g[f"rs{i}_g"].to(dev))
        close(a.grad, g[f"rs{i}_grad"], atol=1e-6, rtol=1e-5, what=f"resize {i} grad")
    hg = _hourglass_from_fixture(g, dev)
    x = g["x"].to(dev).requires_grad_(True)
    out1, pre1, post1 = hg(x)
    out2, _, _ = hg(out1 + x, pre1, post1)
    for name, t in (("out1", out1), ("pre1", pre1), ("post1", post1), ("out2", out2)):
        close(t, g[name], atol=1e-5, rtol=1e-4, what=name)
    (out2 + x).backward(g["g_out"].to(dev))
    close(x.grad, g["grad_x"], atol=1e-5, rtol=1e-4, scale="max", what="hourglass grad_x")
    for k, v in hg.named_parameters():
        close(v.grad, g["gw_" + k], atol=1e-5, rtol=1e-4, scale="max", what="hourglass grad " + k)


@pytest.mark.parametrize("shape,size", [((1, 32, 8, 100, 100), (16, 200, 200)), ((2, 32, 4, 50, 50), (8, 100, 100)),
                                        ((1, 3, 5, 33, 65), (11, 67, 129))])
def test_resize_full_size(dev, shape, size):
    """The UNet's two resize levels at cfg-B size (and a ragged one) against aten on the same GPU:
    forward bit for bit (same expression), backward to summation order."""
    import torch.nn.functional as F
    from vampire_amd.ops import upsample_trilinear
    gen = torch.Generator().manual_seed(9)
    x = torch.randn(shape, generator=gen).to(dev)
    up = torch.randn(shape[:2] + size, generator=gen).to(dev)
    a = x.clone().requires_grad_(True)
    ref = F.interpolate(a, size, mode="trilinear", align_corners=True)
    ref.backward(up)
    b = x.clone().requires_grad_(True)
    got = upsample_trilinear(b, size)
    close(got, ref, atol=1e-6, rtol=1e-6, what="resize")
    got.backward(up)
    close(b.grad, a.grad, atol=1e-5, rtol=1e-5, what="resize grad")
    # adjoint identity <up, R x> == <R^T up, x>
    lhs, rhs = float((up.double() * got.double()).sum()), float((b.grad.double() * x.double()).sum())
    assert abs(lhs - rhs) <= 1e-5 * max(1.0, abs(lhs))


# --------------------------------------------------------------------------- 3x3x3 conv (UNet)
@pytest.mark.parametrize("cin,cout,vol,batch", [(16, 16, (6, 10, 14), 2), (32, 16, (5, 37, 71), 1),
                                                (16, 32, (4, 9, 130), 1), (32, 32, (8, 100, 100), 1),
                                                (16, 16, (16, 200, 200), 1), (32, 16, (4, 12, 256), 1)])
def test_conv3d_matches_torch(dev, cin, cout, vol, batch):
    """the fp32 matrix-core 3x3x3 conv (bv2:20, 40-60 layer shapes, ragged sizes)
    against torch's conv3d evaluated in fp64 on the CPU: output, data and weight gradients."""
    import torch.nn.functional as F
    from vampire_amd.ops import conv3d_3x3x3, conv3d_supported
    assert not conv3d_supported(torch.zeros(1, 32, 2, 2, 400, device=dev), torch.zeros(16, 32, 3, 3, 3, device=dev),
                                (1, 1, 1), (1, 1, 1), None)            # row too long: the module falls back to MIOpen
    gen = torch.Generator().manual_seed(11)
    x = torch.randn(batch, cin, *vol, generator=gen)
    w = torch.randn(cout, cin, 3, 3, 3, generator=gen) * 0.05
    up = torch.randn(batch, cout, *vol, generator=gen)
    a, wa = x.double().requires_grad_(True), w.double().requires_grad_(True)
    F.conv3d(a, wa, padding=1).backward(up.double())
    ref = F.conv3d(x.double(), w.double(), padding=1)
    b, wb = x.to(dev).requires_grad_(True), w.to(dev).requires_grad_(True)
    out = conv3d_3x3x3(b, wb)
    close(out, ref, atol=1e-5, rtol=1e-5, scale="max", what="conv out")
    out.backward(up.to(dev))
    close(b.grad, a.grad, atol=1e-5, rtol=1e-5, scale="max", what="conv grad_in")
    close(wb.grad, wa.grad, atol=1e-5, rtol=2e-5, scale="max", what="conv grad_weight")


def test_module_conv_routing_matches_miopen(dev, monkeypatch):
    """The module mirror's routing of its 3-D convolutions (Unet3D stride-1 layers, the three heads
    fused into one 16 -> 32 conv) through the HIP kernels against the same modules on MIOpen:
    outputs, input gradient and every parameter gradient."""
    from vampire_amd.backbone import BaseVAMPIRE2, Unet3D, SWITCHES
    monkeypatch.setattr(SWITCHES, "conv3d_min_voxels", 0)
    torch.manual_seed(3)
    net = Unet3D(16, 16).to(dev)
    c = CFG_TINY
    mod = BaseVAMPIRE2(x_bound_seg=list(c.x_bound_seg), y_bound_seg=list(c.y_bound_seg), z_bound_seg=list(c.z_bound_seg),
                       x_bound_det=list(c.x_bound_det), y_bound_det=list(c.y_bound_det), z_bound_det=list(c.z_bound_det),
                       d_bound=list(c.d_bound), final_dim=c.final_dim, downsample_factor=4, upsample_factor=4,
                       mid_channels=16, output_channels=8, img_backbone_conf=dict(),
                       img_neck_conf=dict(out_channels=[8] * 4), num_classes=18, density_mode="sdf").to(dev)
    x = torch.randn(2, 16, 8, 24, 40, device=dev)
    up = torch.randn(2, 16, 8, 24, 40, device=dev)
    ups = [torch.randn(2, k, 8, 24, 40, device=dev) for k in (1, 18, 3)]
    heads = [mod.density_conv, mod.seg_conv, mod.rgb_conv]

    def run(flag):
        monkeypatch.setattr(SWITCHES, "conv3d", flag == "1")
        for m in [net] + heads:
            m.zero_grad(set_to_none=True)
        a = x.clone().requires_grad_(True)
        y = net(a)
        outs = mod._heads(y)
        torch.autograd.backward((y,) + tuple(outs), [up] + ups)
        grads = [p.grad.clone() for m in [net] + heads for p in m.parameters()]
        return [y.detach()] + [o.detach() for o in outs], a.grad.clone(), grads

    o1, gx1, gp1 = run("1")
    o0, gx0, gp0 = run("0")
    for i, (a, b) in enumerate(zip(o1, o0)):
        close(a, b, atol=1e-5, rtol=1e-4, scale="max", what=f"module conv routing: output {i}")
    close(gx1, gx0, atol=1e-5, rtol=1e-4, scale="max", what="module conv routing: grad_in")
    names = [n for m in [net] + heads for n, _ in m.named_parameters()]
    for n, a, b in zip(names, gp1, gp0):
        close(a, b, atol=1e-5, rtol=2e-4, scale="max", what="module conv routing: grad " + n)


# --------------------------------------------------------------------------- point resampling
def test_point_resampling_tiny(dev):
    """occupancy and lidar-point queries (bv2:576-609) against the fixture made with
    the reference module's occ_coords / density / bounds: values and all input gradients."""
    g = load_golden("tiny_points.npz")
    cfg = dataclasses.replace(CFG_TINY, density_mode="sdf")
    hp = hot(cfg, dev)
    sem = g["semantic_logits"].to(dev).requires_grad_(True)
    dens = g["density_feature"].to(dev).requires_grad_(True)
    beta = g["beta"].reshape(()).to(dev).requires_grad_(True)
    occ_logits, occ_density = hp.occupancy_queries(sem, dens, g["occ_sub"].to(dev), g["bda"].to(dev), beta)
    pts = g["points"].to(dev)
    pts_logits = hp.sample_points(sem, pts, padding="border", channel_last=True)
    pts_sdf = hp.sample_points(dens, pts, mask_outside=True)[:, 0]
    outs = dict(occ_logits=occ_logits, occ_density=occ_density, pts_logits=pts_logits, pts_sdf=pts_sdf)
    for k, v in outs.items():
        close(v, g[k], atol=1e-5, rtol=1e-5, what=k)
    torch.autograd.backward(list(outs.values()), [g["g_" + k].to(dev) for k in outs])
    close(sem.grad, g["grad_semantic_logits"], atol=1e-5, rtol=1e-5, scale="max", what="grad_semantic_logits")
    close(dens.grad, g["grad_density_feature"], atol=1e-5, rtol=1e-5, scale="max", what="grad_density_feature")
    close(beta.grad.reshape(1), g["grad_beta"], atol=1e-3, rtol=1e-3, what="grad_beta")


def test_occupancy_queries_full_size(dev):
    """The 200x200x16 occupancy grid at cfg-B (rotated by bda) against the oracle on the host,
    forward and the gradient of the semantic volume; plus clustered lidar-like points (many per
    voxel: the heavy-voxel path of the gather)."""
    from oracle import aten_oracle as O
    from vampire_amd.geometry import make_occ_coords
    cfg = CFG_B
    hp = hot(cfg, dev)
    bounds = (cfg.x_bound_seg, cfg.y_bound_seg, cfg.z_bound_seg)
    dens, sem, _, _ = synthetic.render_inputs(cfg, 1, seed=4)
    bda = synthetic.bda_matrix(1, rot_deg=5.0)
    occ = make_occ_coords()
    beta = torch.tensor(0.1)
    # the rotated grid is computed once on the host and handed to both sides: a matmul on the GPU
    # rounds differently, and on white-noise volumes 1e-6 m of coordinate is 1e-4 of value
    rot = bda[:, :3, :3].view(1, 1, 1, 1, 3, 3)
    pts = (rot @ occ[None, ..., None]).squeeze(-1).reshape(1, -1, 3)
    ref_logits = O.sample_points(sem, pts, bounds, "border")
    ref_dens = O.sample_points(O.density_apply(dens, cfg.density_mode, beta, cfg.sdf_bias), pts, bounds)
    sem_d = sem.to(dev).requires_grad_(True)
    got_logits = hp.sample_points(sem_d, pts.to(dev), padding="border")
    got_dens = hp.sample_points(dens.to(dev), pts.to(dev), activation=True, beta=beta.to(dev))
    close(got_logits, ref_logits, atol=1e-5, rtol=1e-5, what="occ_logits")
    close(got_dens, ref_dens, atol=1e-5, rtol=1e-5, what="occ_density")
    gen = torch.Generator().manual_seed(2)
    up = torch.randn(ref_logits.shape, generator=gen)
    sem_c = sem.clone().requires_grad_(True)
    O.sample_points(sem_c, pts, bounds, "border").backward(up)
    got_logits.backward(up.to(dev))
    close(sem_d.grad, sem_c.grad, atol=1e-5, rtol=2e-5, scale="max", what="grad_semantic_logits (occ grid)")
    # 40k points crowded into a 2 m cube: thousands of points per voxel
    pts = (torch.rand(1, 40000, 3, generator=gen) * 2.0 + torch.tensor([3.0, -7.0, 0.5]))
    sem_c.grad = None
    sem_d.grad = None
    upp = torch.randn(1, cfg.num_classes, 40000, generator=gen)
    O.sample_points(sem_c, pts, bounds, "border").backward(upp)
    out = hp.sample_points(sem_d, pts.to(dev), padding="border")
    close(out, O.sample_points(sem, pts, bounds, "border"), atol=1e-5, rtol=1e-5, what="clustered points")
    out.backward(upp.to(dev))
    close(sem_d.grad, sem_c.grad, atol=1e-4, rtol=1e-4, scale="max", what="grad_semantic_logits (clustered)")


# --------------------------------------------------------------------------- full size
def _sha(t):
    return hashlib.sha256(np.ascontiguousarray(t.cpu().numpy()).tobytes()).hexdigest()


@pytest.mark.parametrize("name,cfg", [("A", CFG_A), ("B", CFG_B)])
def test_full_size_checksums(dev, name, cfg):
    """cfg-A / cfg-B at B=1: index tensors hash-equal to the reference's; value statistics and
    probes within tolerance (tests/golden/full_checksums.json)."""
    with open(os.path.join(GOLDEN, "full_checksums.json")) as f:
        ref = json.load(f)[name]
    hp = hot(cfg, dev)
    lm = torch.tensor(ref["lift_mats"], dtype=torch.float32, device=dev)
    rm = torch.tensor(ref["render_mats"], dtype=torch.float32, device=dev)
    # the pinned matrices are what the synthetic rig gives, up to torch.inverse's CPU dependence
    s2e, K, ida = synthetic.camera_rig(cfg, 1)
    torch.testing.assert_close(lift_matrices(s2e, K, ida, synthetic.bda_matrix(1)), lm.cpu(),
                               rtol=1e-5, atol=1e-5)
    valid, ix0, iy0, iz0 = hp.lift_indices(lm)
    assert int(valid.sum()) == ref["lift_valid_count"]
    assert _sha(valid) == ref["lift_valid_sha256"]
    vb = valid.bool()
    for nm, t in (("ix0", ix0), ("iy0", iy0), ("iz0", iz0)):
        assert _sha(torch.where(vb, t, torch.zeros_like(t))) == ref[f"lift_{nm}_sha256"], nm
    inside, rx, ry, rz = hp.render_indices(render_mats=rm)
    assert int(inside.sum()) == ref["render_inside_count"]
    assert _sha(inside) == ref["render_inside_sha256"]
    for nm, t in (("ix0", rx), ("iy0", ry), ("iz0", rz)):
        assert _sha(t) == ref[f"render_{nm}_sha256"], nm

    depth, feat = synthetic.lift_inputs(cfg, 1, seed=0, device=dev)
    vox = hp.lift(depth, feat, lm)
    probe = vox.flatten()[::65537][:64].cpu()
    close(probe, torch.tensor(ref["lift_probe"]), atol=1e-5, what="lift probe")
    assert abs(float(vox.double().abs().sum()) - ref["lift"]["abs_sum"]) <= 1e-5 * ref["lift"]["abs_sum"]

    vols = synthetic.render_inputs(cfg, 1, seed=0, device=dev)
    beta = torch.tensor(0.1, device=dev)
    outs = hp.render(*vols, beta, render_mats=rm)
    close(outs[2].flatten()[::1013][:64].cpu(), torch.tensor(ref["depth_preds_probe"]), what="depth probe")
    close(outs[1].flatten()[::10007][:64].cpu(), torch.tensor(ref["seg_probe"]), what="seg probe")
    for nm, o in zip(NAMES, outs):
        st = ref[nm]
        assert list(o.shape) == st["shape"], nm
        tot = float(o.double().abs().sum())
        assert abs(tot - st["abs_sum"]) <= 2e-5 * st["abs_sum"] + 1e-3, (nm, tot, st["abs_sum"])
        assert abs(float(o.max()) - st["max"]) <= 1e-4 * max(1.0, abs(st["max"])), nm


def test_full_size_properties(dev):
    """Size-independent properties at cfg-B, B=2: linearity of the lift in feat, compositing
    weights bounded by one, batch independence."""
    cfg = CFG_B
    hp = hot(cfg, dev)
    s2e, K, ida = synthetic.camera_rig(cfg, 2, jitter=2.0, seed=3)
    bda = synthetic.bda_matrix(2, rot_deg=5.0)
    lm = lift_matrices(s2e, K, ida, bda).to(dev)
    rm = render_matrices(s2e, K, ida, bda).to(dev)
    depth, feat = synthetic.lift_inputs(cfg, 2, seed=1, device=dev)
    a = hp.lift(depth, feat, lm)
    b = hp.lift(depth, 2.0 * feat, lm)
    close(b, 2.0 * a, atol=1e-6, rtol=1e-5, what="lift linearity in feat")
    # batch independence: sample 1 alone == sample 1 in the batch
    a1 = hp.lift(depth[1:], feat[1:], lm[1:])
    assert torch.equal(a1[0], a[1])
    vols = synthetic.render_inputs(cfg, 2, seed=1, device=dev)
    beta = torch.tensor(0.1, device=dev)
    outs = hp.render(*vols, beta, render_mats=rm)
    d = outs[2]
    assert float(d.min()) >= cfg.d_bound[0] - 1e-3 and float(d.max()) <= cfg.d_bound[1] + 1e-3
    # rgb volume in [0,1] and weights summing to <= 1  =>  rendered rgb in [0,1]
    assert float(outs[0].min()) >= -1e-5 and float(outs[0].max()) <= 1.0 + 1e-5
    o1 = hp.render(*[v[1:] for v in vols], beta, render_mats=rm[1:])
    for x, y in zip(o1, outs):
        assert torch.equal(x[0], y[1])


def test_edge_cases(tiny_common, dev):
    """Degenerate shapes and inputs: one camera and one sample, a camera that sees nothing, all-zero
    upstream gradients, an empty point list, repeated forward calls (bit-identical)."""
    from oracle import aten_oracle as O
    g = tiny_common
    cfg = dataclasses.replace(CFG_TINY, density_mode="sdf")
    hp = hot(cfg, dev)
    geo = PathGeometry(cfg)
    lm, rm = tiny_mats(g, dev)
    # --- B = 1, N = 1 against the oracle (camera 2 of sample 1)
    d1, f1 = g["depth"][1:, 2:3], g["feat"][1:, 2:3]
    ref = O.lift(d1, f1, geo.voxel_coords, None, None, None, None, cfg.final_dim, cfg.d_bound,
                 prepared=g["lift_mats"][1:, 2:3])
    dd, fd = d1.to(dev).requires_grad_(True), f1.to(dev).requires_grad_(True)
    out = hp.lift(dd, fd, lm[1:, 2:3])
    close(out, ref, atol=1e-5, what="lift B=1 N=1")
    out.sum().backward()
    assert bool(torch.isfinite(dd.grad).all()) and bool(torch.isfinite(fd.grad).all())
    # --- a camera looking away from every voxel: identical result to leaving it out
    far = lm.clone()
    far[:, 0, 1, :3, 3] += 1e4                   # push camera 0's projection far off the image
    a = hp.lift(g["depth"].to(dev), g["feat"].to(dev), far)
    valid, *_ = hp.lift_indices(far)
    assert int(valid[:, 0].sum()) == 0
    b = hp.lift(g["depth"][:, 1:].to(dev), g["feat"][:, 1:].to(dev), far[:, 1:])
    assert torch.equal(a, b)
    # --- zero upstream gradients give zero input gradients (and no nan)
    vols = [g[k].to(dev).requires_grad_(True) for k in ("density_feature", "semantic_logits", "base", "rgb")]
    beta = torch.tensor(0.1, device=dev, requires_grad=True)
    outs = hp.render(*vols, beta, render_mats=rm)
    torch.autograd.backward(outs, [torch.zeros_like(o) for o in outs])
    for v in vols:
        assert float(v.grad.abs().max()) == 0.0
    assert float(beta.grad.abs()) == 0.0
    # --- forward is deterministic bit for bit (calls of the same kind, and on the same camera forward: left
    # at "auto" the choice may move between calls with what the rays did -- the two forwards agree to rounding)
    hp.impl["cam_direct"] = True
    outs = hp.render(*vols, beta, render_mats=rm)
    outs2 = hp.render(*[v.detach() for v in vols], beta.detach(), render_mats=rm)
    outs3 = hp.render(*[v.detach() for v in vols], beta.detach(), render_mats=rm)
    for x, y in zip(outs2, outs3):
        assert torch.equal(x, y)
    outs4 = hp.render(*vols, beta, render_mats=rm)
    for x, y in zip(outs, outs4):
        assert torch.equal(x, y)
    for nm, x, y in zip(NAMES, outs, outs2):
        close(x, y, atol=2e-5, rtol=1e-5, what=f"training-mode vs forward-only {nm}")
    # --- empty point list
    sem = g["semantic_logits"].to(dev).requires_grad_(True)
    empty = hp.sample_points(sem, torch.zeros(sem.shape[0], 0, 3, device=dev), padding="border")
    assert empty.shape == (sem.shape[0], sem.shape[1], 0)
    empty.sum().backward()
    assert float(sem.grad.abs().max()) == 0.0


@pytest.mark.parametrize("cfg", [CFG_B, CFG_A], ids=["B", "A"])
def test_camera_backward_cell_matches_atomic_splat_full_size(dev, monkeypatch, cfg):
    """The default camera backward (per-ray pass, samples sorted into voxel cells, per-voxel
    gather; one- and two-stream form) against the v1 float-atomic splat at cfg-B: two independent
    HIP implementations of the same gradient (cfg-B and the reference's default cfg-A)."""
    hp = hot(cfg, dev)
    s2e, K, ida = synthetic.camera_rig(cfg, 1, jitter=2.0, seed=5)
    bda = synthetic.bda_matrix(1, rot_deg=7.0, flip_dy=True)
    rm = render_matrices(s2e, K, ida, bda).to(dev)
    beta = torch.tensor(0.1, device=dev, requires_grad=True)
    gen = torch.Generator(device=dev).manual_seed(9)

    def run(impl):
        hp.impl["cam_bwd"] = impl
        vols = [v.requires_grad_(True) for v in synthetic.render_inputs(cfg, 1, seed=2, device=dev)]
        beta.grad = None
        outs = hp.render(*vols, beta, render_mats=rm)
        gen.manual_seed(9)
        gs = [torch.randn(o.shape, device=dev, generator=gen) for o in outs]
        for i in (3, 4, 5, 6, 7):          # camera branch only: zero the BEV upstream grads
            gs[i].zero_()
        torch.autograd.backward(outs, gs)
        return [v.grad.clone() for v in vols], beta.grad.clone()

    g4, b4 = run("cell")        # the default: the per-ray pass reads the samples the forward kept (save_rows)
    hp.impl["save_rows"] = False
    g3, b3 = run("cell")        # the per-ray pass gathers again: BEV branch on the side stream, camera gather accumulates
    hp.impl["overlap"] = False
    g2, b2 = run("cell")        # same kernels on one stream, camera gather overwrites
    hp.impl["overlap"] = True
    hp.impl["save_rows"] = True
    g1, b1 = run("v1")          # float-atomic splat
    for name, a4, a3, a2, b in zip(("density_feature", "semantic_logits", "base", "rgb"), g4, g3, g2, g1):
        close(a3, b, atol=1e-5, rtol=2e-5, scale="max", what="cell (2 streams) vs v1 grad_" + name)
        close(a2, b, atol=1e-5, rtol=2e-5, scale="max", what="cell (1 stream) vs v1 grad_" + name)
        # the kept samples were taken at the one-kernel forward's coordinates (an fp64 line per ray, within
        # 1e-5 voxel of the fp32 chain v1 evaluates): <= 3e-5 of the sample values on white noise, which the
        # density gradient sees through q = G . s; both are pinned to the reference at 1e-4 elsewhere
        close(a4, b, atol=1e-5, rtol=6e-5, scale="max", what="cell (kept samples) vs v1 grad_" + name)
        assert float(b.abs().max()) > 0 or name == "base"
    for bb in (b4, b3, b2):
        close(bb.reshape(1), b1.reshape(1), atol=1e-3, rtol=1e-3, what="grad_beta")


@pytest.mark.parametrize("mode,cat_seg,batch", [("naive", True, 2), ("sdf", True, 1), ("naive", False, 1)],
                         ids=["naive-catseg-b2", "sdf-catseg", "naive"])
def test_render_variants_full_size_match_atomic_splat(dev, monkeypatch, mode, cat_seg, batch):
    """cfg-B with the other density mode / cat_seg / batch 2 (per-sample camera rigs): all eight
    render outputs are finite and every input gradient of the default backward (cell lists, BEV
    gathers, two streams) equals the v1 float-atomic implementations'."""
    cfg = dataclasses.replace(CFG_B, density_mode=mode, cat_seg=cat_seg)
    hp = hot(cfg, dev)
    s2e, K, ida = synthetic.camera_rig(cfg, batch, jitter=2.0, seed=7)
    bda = synthetic.bda_matrix(batch, rot_deg=-4.0)
    rm = render_matrices(s2e, K, ida, bda).to(dev)
    beta = torch.tensor(0.1, device=dev, requires_grad=True) if mode == "sdf" else None
    gen = torch.Generator(device=dev).manual_seed(3)

    def run(impl):
        hp.impl["cam_bwd"] = hp.impl["bev_bwd"] = impl
        vols = [v.requires_grad_(True) for v in synthetic.render_inputs(cfg, batch, seed=6, device=dev)]
        if beta is not None:
            beta.grad = None
        outs = hp.render(*vols, beta, render_mats=rm)
        gen.manual_seed(3)
        gs = [torch.randn(o.shape, device=dev, generator=gen) for o in outs]
        torch.autograd.backward(outs, gs)
        return outs, [v.grad.clone() for v in vols], (beta.grad.clone() if beta is not None else None)

    outs, g_new, b_new = run("cell")
    _, g_old, b_old = run("v1")
    assert outs[7].shape[1] == cfg.mid_channels + (cfg.num_classes if cat_seg else 0)
    for name, o in zip(NAMES, outs):
        assert bool(torch.isfinite(o).all()), name
    for name, a, b in zip(("density_feature", "semantic_logits", "base", "rgb"), g_new, g_old):
        assert float(b.abs().max()) > 0, name
        # (4e-5: with the kept samples the per-ray pass works on the one-kernel forward's line coordinates, the v1
        # splat on the fp32 chain's -- 3.1e-5 of the largest gradient without early termination, VAMP_ERT=0)
        close(a, b, atol=1e-5, rtol=4e-5, scale="max", what=f"{mode} cat_seg={cat_seg}: grad_" + name)
    if beta is not None:
        close(b_new.reshape(1), b_old.reshape(1), atol=1e-3, rtol=1e-3, what="grad_beta")


def test_bev_backward_gather_matches_atomic_splat_full_size(dev, monkeypatch):
    """BEV-branch backward: per-voxel gather (v2) against the column-thread atomic splat (v1)."""
    import dataclasses
    cfg = dataclasses.replace(CFG_B, cat_seg=True)
    hp = hot(cfg, dev)
    s2e, K, ida = synthetic.camera_rig(cfg, 1)
    rm = render_matrices(s2e, K, ida, synthetic.bda_matrix(1)).to(dev)
    beta = torch.tensor(0.1, device=dev, requires_grad=True)
    gen = torch.Generator(device=dev)

    def run(impl):
        hp.impl["bev_bwd"] = impl
        vols = [v.requires_grad_(True) for v in synthetic.render_inputs(cfg, 1, seed=4, device=dev)]
        beta.grad = None
        outs = hp.render(*vols, beta, render_mats=rm)
        gen.manual_seed(21)
        gs = [torch.randn(o.shape, device=dev, generator=gen) for o in outs]
        for i in (0, 1, 2):                # BEV branch only: zero the camera upstream grads
            gs[i].zero_()
        torch.autograd.backward(outs, gs)
        return [v.grad.clone() for v in vols], beta.grad.clone()

    g2, b2 = run("cell")
    g1, b1 = run("v1")
    for name, a, b in zip(("density_feature", "semantic_logits", "base", "rgb"), g2, g1):
        close(a, b, atol=1e-5, rtol=2e-5, scale="max", what="bev v2 vs v1 grad_" + name)
        assert float(b.abs().max()) > 0
    close(b2.reshape(1), b1.reshape(1), atol=1e-3, rtol=1e-3, what="grad_beta")


@pytest.mark.parametrize("accumulate", [False, True], ids=["overwrite", "accumulate"])
def test_bev_backward_round4_kernels_edge_shapes(dev, accumulate):
    """The round-4 BEV backward kernels where cfg-A / B / D do not take them: 26 classes (11 channel groups for the
    8 waves of bev_gather_comp_kernel: a second round of groups; 29 composited channels: the any-channel-count
    path of bev_qscan_saved_kernel), 13 lattice heights (two chunks of taps; 13 waves in the q + scan), a 40 x 40
    grid whose last block of 64 columns is ragged -- against the float-atomic splat (v1), with the BEV branch
    overwriting (default) and adding on top of the camera branch's gradient (one stream behind the v1 camera splat)."""
    cfg = dataclasses.replace(CFG_B, x_bound_seg=(-8.0, 8.0, 0.4), y_bound_seg=(-8.0, 8.0, 0.4),
                              x_bound_det=(-8.0, 8.0, 0.4), y_bound_det=(-8.0, 8.0, 0.4),
                              z_bound_det=(-1.0, 3.0, 0.3), num_classes=26, final_dim=(64, 176))
    assert cfg.oZ == 13 and cfg.oX == 40
    hp = hot(cfg, dev)
    if accumulate:
        hp.impl["overlap"] = False
        hp.impl["cam_bwd"] = "v1"          # the camera splat first, the BEV kernels add on top (their accumulate mode)
    s2e, K, ida = synthetic.camera_rig(cfg, 2, jitter=1.0, seed=5)
    rm = render_matrices(s2e, K, ida, synthetic.bda_matrix(2)).to(dev)
    beta = torch.tensor(0.1, device=dev, requires_grad=True)
    gen = torch.Generator(device=dev)

    def run(impl):
        hp.impl["bev_bwd"] = impl
        vols = [v.requires_grad_(True) for v in synthetic.render_inputs(cfg, 2, seed=4, device=dev)]
        beta.grad = None
        outs = hp.render(*vols, beta, render_mats=rm)
        gen.manual_seed(21)
        gs = [torch.randn(o.shape, device=dev, generator=gen) for o in outs]
        torch.autograd.backward(outs, gs)
        return [v.grad.clone() for v in vols], beta.grad.clone()

    g2, b2 = run("cell")
    g1, b1 = run("v1")
    for name, a, b in zip(("density_feature", "semantic_logits", "base", "rgb"), g2, g1):
        assert float(b.abs().max()) > 0
        close(a, b, atol=1e-5, rtol=3e-5, scale="max", what="bev edge shapes: grad_" + name)
    close(b2.reshape(1), b1.reshape(1), atol=1e-3, rtol=1e-3, what="grad_beta")


@pytest.mark.parametrize("cfg", [CFG_B, CFG_A], ids=["B", "A"])
def test_lift_backward_cell_matches_atomic_splat_full_size(dev, monkeypatch, cfg):
    """Lift backward: cell list + a wave per pixel (default; also with 4 and 16 waves per pixel,
    the dense-pixel configurations) against the per-voxel atomic splat (v1), cfg-B and cfg-A,
    B=2 with jittered rigs and a bda rotation."""
    hp = hot(cfg, dev)
    s2e, K, ida = synthetic.camera_rig(cfg, 2, jitter=2.0, seed=11)
    bda = synthetic.bda_matrix(2, rot_deg=-6.0, scale=1.02)
    lm = lift_matrices(s2e, K, ida, bda).to(dev)
    gen = torch.Generator(device=dev)

    def run(impl):
        hp.impl["lift_bwd"] = impl
        depth, feat = synthetic.lift_inputs(cfg, 2, seed=6, device=dev)
        depth.requires_grad_(True); feat.requires_grad_(True)
        out = hp.lift(depth, feat, lm)
        gen.manual_seed(33)
        out.backward(torch.randn(out.shape, device=dev, generator=gen))
        return depth.grad.clone(), feat.grad.clone()

    d4, f4 = run("cell")        # default: cell list + a wave per pixel
    hp.impl["lift_wpp"] = 4
    d5, f5 = run("cell")        # same, four waves per pixel (the dense-pixel configuration)
    hp.impl["lift_wpp"] = 16
    d6, f6 = run("cell")
    hp.impl["lift_wpp"] = 0
    d1, f1 = run("v1")          # per-voxel float-atomic splat
    assert float(d1.abs().max()) > 0 and float(f1.abs().max()) > 0
    for tag, dd, ff in (("cell", d4, f4), ("cell wpp4", d5, f5), ("cell wpp16", d6, f6)):
        close(dd, d1, atol=1e-6, rtol=2e-5, scale="max", what=tag + " vs splat grad_depth")
        close(ff, f1, atol=1e-6, rtol=2e-5, scale="max", chan_dim=2, what=tag + " vs splat grad_feat")


def test_backbone_forward_backward(dev):
    """The drop-in BaseVAMPIRE2 module end to end on the GPU against the same module with the lift
    and the renderer swapped for the oracle on CPU (same weights, same inputs): all 12 outputs."""
    import copy
    from oracle import aten_oracle as O
    from vampire_amd.backbone import BaseVAMPIRE2
    c = CFG_TINY
    kw = dict(x_bound_seg=list(c.x_bound_seg), y_bound_seg=list(c.y_bound_seg), z_bound_seg=list(c.z_bound_seg),
              x_bound_det=list(c.x_bound_det), y_bound_det=list(c.y_bound_det), z_bound_det=list(c.z_bound_det),
              d_bound=list(c.d_bound), final_dim=c.final_dim, downsample_factor=4, upsample_factor=4,
              mid_channels=4, output_channels=8, img_backbone_conf=dict(), img_neck_conf=dict(out_channels=[8] * 4),
              num_classes=5, density_mode="sdf", sdf_bias=-1.0, cat_pos=True, cat_seg=True)
    torch.manual_seed(0)
    ref = BaseVAMPIRE2(**kw).eval()
    with torch.no_grad():
        ref.density_conv.bias.fill_(-1.0)          # bring densities into an informative range
    mod = copy.deepcopy(ref).to(dev)
    B = 2
    s2e, K, ida = synthetic.camera_rig(c, B, src_hw=(64, 176), focal=60.0, centre=(88.0, 34.0), jitter=2.0, seed=4)
    s2e[:, :, :3, 3] *= 0.3
    bda = synthetic.bda_matrix(B, rot_deg=8.0)
    mats = dict(sensor2ego_mats=s2e[:, None], intrin_mats=K[:, None], ida_mats=ida[:, None],
                sensor2sensor_mats=torch.eye(4).expand(B, 1, 6, 4, 4), bda_mat=bda)
    imgs = torch.randn(B, 1, 6, 3, *c.final_dim)
    pts = [torch.rand(50, 3) * 8 - 4 for _ in range(B)]
    imgs_d = imgs.to(dev).requires_grad_(True)
    out = mod(imgs_d, {k: v.to(dev) for k, v in mats.items()}, inrange_pts=[p.to(dev) for p in pts])
    assert len(out) == 12

    # the same forward with oracle lift/render on CPU
    def oracle_forward(m, imgs):
        feats = m.get_cam_feats(imgs)
        src = feats[:, 0].reshape(B * 6, -1, feats.shape[-2], feats.shape[-1])
        depth = m.mapping_along_depth(src).softmax(dim=1).reshape(B, 6, -1, *src.shape[-2:])
        feat = m.channel_lower(src).reshape(B, 6, -1, *src.shape[-2:])
        vox = O.lift(depth, feat, m.voxel_coords, s2e, K, ida, bda, c.final_dim, c.d_bound)
        vox = torch.cat([vox, m.norm_voxel_coords.permute(3, 0, 1, 2)[None].repeat(B, 1, 1, 1, 1)], 1)
        base = m.base_conv(vox)
        dens, sem, rgb = m.density_conv(base), m.seg_conv(base), m.rgb_conv(base)
        geom = torch.nan_to_num(O.frustum_to_ego(m.frustum, s2e, K, ida, bda), -1e3)
        r = O.render(geom, dens, sem, base, rgb,
                     seg_bounds=(c.x_bound_seg, c.y_bound_seg, c.z_bound_seg),
                     output_coords=m.output_coords, camera_mids=m.camera_mids, bev_mids=m.bev_mids,
                     d_far=c.d_bound[1], z_step_det=c.z_bound_det[2], num_classes=5, density_mode="sdf",
                     beta_param=m.density.beta, sdf_bias=-1.0, cat_seg=True)
        return vox, r, dens, sem

    imgs_c = imgs.clone().requires_grad_(True)
    vox_o, r, dens_o, sem_o = oracle_forward(ref, imgs_c)
    up = lambda t: ref.upsample2d(t.reshape(B * 6, -1, ref.fH, ref.fW)).reshape(B, 6, -1, ref.fH * 4, ref.fW * 4)
    close(out[1], up(r[0]), what="rgb_preds")
    close(out[2], up(r[1]), what="seg_logits_preds")
    close(out[3], up(r[2]), atol=2e-4, what="depth_preds")
    for i, j in ((4, 3), (5, 4), (6, 5), (7, 6)):
        close(out[i], r[j], what=f"output {i}")
    bev_feat = ref.voxel_output((r[7] * r[6].tanh()).reshape(B, -1, *r[7].shape[-2:]))
    close(out[0], bev_feat, atol=2e-4, what="bev feature")
    assert len(out[8]) == B and out[8][0].shape == (50, 5) and out[10].shape == (B, 200, 200, 16, 5)
    # lidar-point and occupancy queries (bv2:576-609)
    bounds = (c.x_bound_seg, c.y_bound_seg, c.z_bound_seg)
    for i in range(B):
        close(out[8][i], O.sample_points(sem_o[[i]], pts[i][None], bounds, "border")[0].T, atol=2e-4,
              what="pts_logits")
        close(out[9][i], O.sample_points(dens_o[[i]], pts[i][None], bounds, "zeros", True)[0, 0], atol=2e-4,
              what="pts_sdf")
    ol, od = O.occupancy_queries(sem_o, dens_o, ref.occ_coords, bda, bounds, "sdf", ref.density.beta, -1.0)
    close(out[10], ol.permute(0, 2, 3, 4, 1), atol=2e-4, what="occ_logits")
    close(out[11], od.permute(0, 2, 3, 4, 1).tanh(), atol=2e-4, what="occ_density")

    # backward through everything: gradient w.r.t. the images and beta
    loss = sum((o.float() ** 2).mean() for o in out[:8])
    loss.backward()
    loss_o = (bev_feat ** 2).mean() + sum((t ** 2).mean() for t in (up(r[0]), up(r[1]), up(r[2]), r[3], r[4], r[5], r[6]))
    loss_o.backward()
    close(imgs_d.grad, imgs_c.grad, atol=1e-6, rtol=2e-3, scale="max", what="grad images")
    close(mod.density.beta.grad.reshape(1), ref.density.beta.grad.reshape(1), atol=1e-5, rtol=2e-3, what="grad beta")


# --------------------------------------------------------------------------- round-2 parity pins
def _block_check(t, ref, what, rtol=2e-5, probe_atol=None, elem_atol=0.0):
    """A tensor against make_golden.block_stat(): 256 contiguous block sums (relative to the
    block's abs-sum: fp32 summation order differs), 64 strided probes, extrema, shape."""
    f = t.detach().double().flatten().cpu()
    n = f.numel()
    assert list(t.shape) == ref["shape"], what
    nb = len(ref["block_sum"])
    edges = [(n * i) // nb for i in range(nb + 1)]
    top = max(abs(ref["max"]), abs(ref["min"]))
    for i in range(nb):
        blk = f[edges[i]:edges[i + 1]]
        lim = rtol * ref["block_abs_sum"][i] + 1e-7 * top * max(1, blk.numel()) ** 0.5 + elem_atol * blk.numel()
        assert abs(float(blk.sum()) - ref["block_sum"][i]) <= lim, \
            f"{what}: block {i} sum {float(blk.sum()):.9e} vs {ref['block_sum'][i]:.9e} (lim {lim:.2e})"
        assert abs(float(blk.abs().sum()) - ref["block_abs_sum"][i]) <= rtol * ref["block_abs_sum"][i] + lim, \
            f"{what}: block {i} abs-sum"
    probe = f[::ref["probe_stride"]][:len(ref["probe"])]
    want = torch.tensor(ref["probe"], dtype=torch.float64)
    atol = ((1e-5 * top) if probe_atol is None else probe_atol) + elem_atol
    assert bool(((probe - want).abs() <= atol + 1e-4 * want.abs()).all()), \
        f"{what}: probes, max err {float((probe - want).abs().max()):.3e} (atol {atol:.2e})"
    assert (abs(float(f.max()) - ref["max"]) <= 1e-4 * top + 1e-7 + elem_atol
            and abs(float(f.min()) - ref["min"]) <= 1e-4 * top + 1e-7 + elem_atol), what


def _upstream(shapes, seed, dev):
    """tests/golden/make_golden.py: upstream_grads (same CPU generator, same order)."""
    g = torch.Generator().manual_seed(seed)
    return [(torch.randn(s, generator=g) * 1e-3).to(dev) for s in shapes]


@pytest.mark.parametrize("impl", ["default", "v1"])
@pytest.mark.parametrize("name,cfg", [("A", CFG_A), ("B", CFG_B)])
def test_full_size_gradients_match_reference(dev, name, cfg, impl):
    """cfg-A / cfg-B at B=1: every gradient of the HIP backward -- the default cell-list path AND the
    v1 float-atomic cross-check path -- against the REFERENCE run
    with autograd on the same seeded inputs and upstream gradients
    (tests/golden/full_grad_checksums.json)."""
    with open(os.path.join(GOLDEN, "full_grad_checksums.json")) as f:
        ref = json.load(f)[name]
    with open(os.path.join(GOLDEN, "full_checksums.json")) as f:
        mats = json.load(f)[name]
    hp = hot(cfg, dev)
    if impl != "default":
        hp.impl["cam_bwd"] = hp.impl["lift_bwd"] = hp.impl["bev_bwd"] = impl
    lm = torch.tensor(mats["lift_mats"], dtype=torch.float32, device=dev)
    rm = torch.tensor(mats["render_mats"], dtype=torch.float32, device=dev)
    depth, feat = synthetic.lift_inputs(cfg, 1, seed=0, device=dev)
    depth.requires_grad_(True); feat.requires_grad_(True)
    vox = hp.lift(depth, feat, lm)
    g_vox = _upstream([vox.shape], ref["seed_lift"], dev)[0]
    # entries where the reference's exact-zero hit test is ill-conditioned (make_golden.py)
    g_vox.view(-1)[ref["lift_upstream_zero_idx"]] = 0.0
    vox.backward(g_vox)
    _block_check(depth.grad, ref["grad_depth"], f"cfg-{name} {impl} grad_depth")
    _block_check(feat.grad, ref["grad_feat"], f"cfg-{name} {impl} grad_feat")
    vols = [v.requires_grad_(True) for v in synthetic.render_inputs(cfg, 1, seed=0, device=dev)]
    beta = torch.tensor(ref["beta"], device=dev, requires_grad=True)
    outs = hp.render(*vols, beta, render_mats=rm)
    torch.autograd.backward(outs, _upstream([o.shape for o in outs], ref["seed_render"], dev))
    for k, v in zip(("density_feature", "semantic_logits", "base", "rgb"), vols):
        _block_check(v.grad, ref["grad_" + k], f"cfg-{name} {impl} grad_{k}", rtol=5e-5)
    assert abs(float(beta.grad) - ref["grad_beta"]) <= 2e-3 * abs(ref["grad_beta"]) + 1e-4, \
        (float(beta.grad), ref["grad_beta"])


@pytest.mark.parametrize("mode,cat_seg", RENDER_VARIANTS)
def test_v1_cross_check_paths_tiny_golden(tiny_common, dev, mode, cat_seg):
    """The v1 float-atomic backward implementations (the cross-check of the full-size tests) are
    themselves pinned against the reference's golden vectors."""
    g = tiny_common
    r = load_golden(render_fixture_name(mode, cat_seg))
    cfg = dataclasses.replace(CFG_TINY, density_mode=mode, cat_seg=cat_seg)
    hp = hot(cfg, dev)
    hp.impl["cam_bwd"] = hp.impl["lift_bwd"] = hp.impl["bev_bwd"] = "v1"
    lm, rm = tiny_mats(g, dev)
    vols = [g[k].to(dev).requires_grad_(True)
            for k in ("density_feature", "semantic_logits", "base", "rgb")]
    beta = (r["beta"].reshape(()).to(dev).requires_grad_(True) if mode == "sdf" else None)
    outs = hp.render(*vols, beta, render_mats=rm)
    torch.autograd.backward(outs, [r["g_" + n].to(dev) for n in NAMES])
    for k, v in zip(("density_feature", "semantic_logits", "base", "rgb"), vols):
        close(v.grad, r["grad_" + k], atol=1e-5, rtol=1e-5, scale="max", what="v1 grad_" + k)
    if mode == "sdf":
        close(beta.grad.reshape(1), r["grad_beta"], atol=1e-3, rtol=1e-3, what="v1 grad_beta")
    d = g["depth"].to(dev).requires_grad_(True)
    f = g["feat"].to(dev).requires_grad_(True)
    hp.lift(d, f, lm).backward(g["g_lift"].to(dev))
    close(d.grad, g["grad_depth"], atol=1e-5, rtol=1e-5, scale="max", what="v1 grad_depth")
    close(f.grad, g["grad_feat"], atol=1e-5, rtol=1e-5, scale="max", chan_dim=2, what="v1 grad_feat")


@pytest.mark.parametrize("cfg,batch,mode", [(CFG_TINY, 2, "sdf"), (CFG_TINY, 2, "naive"), (CFG_B, 1, "sdf")],
                         ids=["tiny-sdf", "tiny-naive", "cfg-B"])
def test_bf16_inputs_equal_fp32_path_on_rounded_values(tiny_common, dev, cfg, batch, mode):
    """bf16 volumes / depth / feat are promoted to fp32 in-kernel: render forward,
    render backward, lift backward and the point queries must give what the fp32 path gives on the
    bf16-rounded values (forward bit for bit; gradients up to the final rounding to bf16)."""
    cfg = dataclasses.replace(cfg, density_mode=mode, cat_seg=(mode == "naive"))
    hp = hot(cfg, dev)
    if cfg.vX == CFG_TINY.vX:
        lm, rm = tiny_mats(tiny_common, dev)
        srcs = [tiny_common[k] for k in ("density_feature", "semantic_logits", "base", "rgb")]
        depth, feat = tiny_common["depth"], tiny_common["feat"]
    else:
        s2e, K, ida = synthetic.camera_rig(cfg, batch)
        bda = synthetic.bda_matrix(batch)
        lm, rm = lift_matrices(s2e, K, ida, bda).to(dev), render_matrices(s2e, K, ida, bda).to(dev)
        srcs = synthetic.render_inputs(cfg, batch, seed=3)
        depth, feat = synthetic.lift_inputs(cfg, batch, seed=3)
    beta = torch.tensor(0.1, device=dev) if mode == "sdf" else None

    def run(dtype):
        vols = [s.to(dev).bfloat16().to(dtype).requires_grad_(True) for s in srcs]
        b = beta.clone().requires_grad_(True) if beta is not None else None
        outs = hp.render(*vols, b, render_mats=rm)
        gen = torch.Generator(device=dev).manual_seed(17)
        torch.autograd.backward(outs, [torch.randn(o.shape, device=dev, generator=gen) for o in outs])
        d = depth.to(dev).bfloat16().to(dtype).requires_grad_(True)
        f = feat.to(dev).bfloat16().to(dtype).requires_grad_(True)
        vox = hp.lift(d, f, lm)
        gen.manual_seed(18)
        vox.backward(torch.randn(vox.shape, device=dev, generator=gen))
        pts = (torch.rand(batch, 500, 3, generator=torch.Generator().manual_seed(2)) * 1.2 - 0.1)
        lo = torch.tensor([cfg.x_bound_seg[0], cfg.y_bound_seg[0], cfg.z_bound_seg[0]])
        hi = torch.tensor([cfg.x_bound_seg[1], cfg.y_bound_seg[1], cfg.z_bound_seg[1]])
        pts = (pts * (hi - lo) + lo).to(dev)
        sem = vols[1].detach().clone().requires_grad_(True)
        q = hp.sample_points(sem, pts, padding="border")
        gen.manual_seed(19)
        q.backward(torch.randn(q.shape, device=dev, generator=gen))
        return (outs, vox, q), [v.grad for v in vols] + [d.grad, f.grad, sem.grad], (b.grad if b is not None else None)

    # (bit-for-bit comparisons of two calls: the camera forward is pinned -- left at "auto" it may move
    # between the calls with what the rays did; the tiny cases take both forwards in turn)
    for direct in ((True, False) if cfg.vX == CFG_TINY.vX else (True,)):
        hp.impl["cam_direct"] = direct
        _bf16_vs_fp32(run)


def _bf16_vs_fp32(run):
    (o16, v16, q16), g16, b16 = run(torch.bfloat16)
    (o32, v32, q32), g32, b32 = run(torch.float32)
    for n, a, b in zip(NAMES, o16, o32):
        assert a.dtype == torch.float32 and torch.equal(a, b), n
    assert torch.equal(v16, v32) and torch.equal(q16, q32)
    for n, a, b in zip(("density_feature", "semantic_logits", "base", "rgb", "depth", "feat", "sem(points)"), g16, g32):
        assert a.dtype == torch.bfloat16, n
        # the fp32 gradient is computed identically up to the order of its sums, then rounded to bf16
        close(a.float(), b, atol=1e-6, rtol=2.0 ** -7, what="bf16 grad_" + n)
        close(a.float(), b, atol=1e-6, rtol=2.0 ** -8, scale="max", what="bf16 grad_" + n + " (max-scaled)")
    if b16 is not None:
        close(b16.reshape(1), b32.reshape(1), atol=1e-3, rtol=1e-3, what="bf16 grad_beta")


def test_cfg_d_bf16_matches_reference(dev):
    """BASELINE configs[3]: 512x1408 input, 400x400x32 grid, bf16 inputs, B=1.  Tap indices and masks
    hash-equal to the reference's (30.7 M lift taps, 23 M ray samples); lift output and all eight
    render outputs against the reference's block sums / probes on the bf16-rounded inputs
    (tests/golden/cfgd_checksums.json); then fwd+bwd properties at full size."""
    from vampire_amd.config import CFG_D as cfg
    with open(os.path.join(GOLDEN, "cfgd_checksums.json")) as f:
        ref = json.load(f)["D"]
    hp = hot(cfg, dev)
    lm = torch.tensor(ref["lift_mats"], dtype=torch.float32, device=dev)
    rm = torch.tensor(ref["render_mats"], dtype=torch.float32, device=dev)
    valid, ix0, iy0, iz0 = hp.lift_indices(lm)
    assert int(valid.sum()) == ref["lift_valid_count"] and _sha(valid) == ref["lift_valid_sha256"]
    vb = valid.bool()
    for nm, t in (("ix0", ix0), ("iy0", iy0), ("iz0", iz0)):
        assert _sha(torch.where(vb, t, torch.zeros_like(t))) == ref[f"lift_{nm}_sha256"], nm
    del valid, ix0, iy0, iz0, vb
    inside, rx, ry, rz = hp.render_indices(render_mats=rm)
    assert int(inside.sum()) == ref["render_inside_count"] and _sha(inside) == ref["render_inside_sha256"]
    for nm, t in (("ix0", rx), ("iy0", ry), ("iz0", rz)):
        assert _sha(t) == ref[f"render_{nm}_sha256"], nm
    del inside, rx, ry, rz
    depth, feat = (t.to(dev).bfloat16().requires_grad_(True) for t in synthetic.lift_inputs(cfg, 1, seed=0))
    vox = hp.lift(depth, feat, lm)
    _block_check(vox, ref["lift"], "cfg-D lift", rtol=2e-5, probe_atol=1e-5)
    vols = [t.to(dev).bfloat16().requires_grad_(True) for t in synthetic.render_inputs(cfg, 1, seed=0)]
    beta = torch.tensor(0.1, device=dev, requires_grad=True)
    outs = hp.render(*vols, beta, render_mats=rm)
    for nm, o in zip(NAMES, outs):
        _block_check(o, ref[nm], "cfg-D " + nm, rtol=3e-5, probe_atol=ATOL)
    # element by element (round 6; make_golden.py --cfgd-samples): 10 000 strided elements of the lift output and of
    # each of the eight rendered tensors from the reference run at this size, 1e-4 ABSOLUTE (north_star's bar)
    es = load_golden("cfgd_samples.npz")
    worst_out = {"lift": _sample_check(vox, es, "lift", "cfg-D lift", abs_tol=1e-4)}
    for nm, o in zip(NAMES, outs):
        worst_out[nm] = _sample_check(o, es, nm, "cfg-D " + nm, abs_tol=1e-4)
    print("cfg-D worst ABSOLUTE output element error:", {k: f"{v[1]:.1e}" for k, v in worst_out.items()})
    # backward at full size: finite, and the adjoint identity of the lift (linear in feat):
    # <g, lift(depth, feat)> == <grad_feat, feat>
    gen = torch.Generator(device=dev).manual_seed(5)
    g = torch.randn(vox.shape, device=dev, generator=gen)
    vox.backward(g)
    lhs = float((g.double() * vox.double()).sum())
    f32 = feat.detach().float().requires_grad_(True)
    vox32 = hp.lift(depth.detach().float(), f32, lm)
    vox32.backward(g)
    rhs = float((f32.grad.double() * f32.detach().double()).sum())
    assert abs(lhs - rhs) <= 1e-4 * max(1.0, abs(lhs)), (lhs, rhs)
    close(feat.grad.float(), f32.grad, atol=1e-6, rtol=2.0 ** -8, scale="max", what="cfg-D bf16 grad_feat")
    torch.autograd.backward(outs, [torch.randn(o.shape, device=dev, generator=gen) * 1e-3 for o in outs])
    for v in vols:
        assert bool(torch.isfinite(v.grad.float()).all()) and float(v.grad.float().abs().max()) > 0
    assert bool(torch.isfinite(beta.grad))
    del vox, vox32, outs, vols, depth, feat, f32, g

    # the six input gradients against the REFERENCE's autograd at this size (bv2:507-514, 419-440 run on the
    # bf16-rounded inputs, tests/golden/make_golden.py --cfgd-grads): 10 000 strided elements of each within
    # 1e-4 of the tensor's scale, grad_beta within 2e-3.  (fp32 tensors holding the bf16-rounded values: the
    # 2-byte input path returns these gradients rounded to bf16, test_bf16_inputs_equal_fp32_path_on_rounded_values)
    gs = load_golden("cfgd_grad_samples.npz")
    rnd = lambda t: t.to(dev).bfloat16().float().requires_grad_(True)
    depth, feat = (rnd(t) for t in synthetic.lift_inputs(cfg, 1, seed=0))
    vox = hp.lift(depth, feat, lm)
    g_vox = _upstream([vox.shape], int(gs["seed_lift"]), dev)[0]
    g_vox.view(-1)[gs["lift_upstream_zero_idx"].to(dev)] = 0.0
    vox.backward(g_vox)
    worst = {"grad_depth": _sample_check(depth.grad, gs, "grad_depth", "cfg-D grad_depth"),
             "grad_feat": _sample_check(feat.grad, gs, "grad_feat", "cfg-D grad_feat")}
    del vox, g_vox, depth, feat
    vols = [rnd(t) for t in synthetic.render_inputs(cfg, 1, seed=0)]
    beta = torch.tensor(float(gs["beta"]), device=dev, requires_grad=True)
    outs = hp.render(*vols, beta, render_mats=rm)
    torch.autograd.backward(outs, _upstream([o.shape for o in outs], int(gs["seed_render"]), dev))
    for k, v in zip(("density_feature", "semantic_logits", "base", "rgb"), vols):
        worst["grad_" + k] = _sample_check(v.grad, gs, "grad_" + k, "cfg-D grad_" + k)
    gb = float(gs["grad_beta"])
    assert abs(float(beta.grad) - gb) <= 2e-3 * abs(gb) + 1e-4, (float(beta.grad), gb)
    print("cfg-D worst gradient element error / scale:", {k: f"{v[0]:.1e}" for k, v in worst.items()},
          "grad_beta", float(beta.grad), "reference", gb)


def test_nccl_world_size_one_smoke(dev):
    """The multi-GPU wiring on the real device with RCCL (backend "nccl"), world size 1: process
    group init, GradSync hook + finish, max_over_ranks, barrier."""
    import torch.distributed as tdist
    from vampire_amd import dist as vdist
    from vampire_amd.step import LiftRenderStep, SyntheticBatch, train_step
    if tdist.is_initialized():
        pytest.skip("a process group already exists in this process")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    try:
        vdist.init("nccl", dev, force=True)
        assert tdist.get_backend() == "nccl" and tdist.get_world_size() == 1
        cfg = dataclasses.replace(CFG_TINY, density_mode="sdf")
        model = LiftRenderStep(cfg, dev)
        wrapped = vdist.wrap_ddp(model, dev, force=True)
        assert isinstance(wrapped, vdist.GradSync)
        data = SyntheticBatch(cfg, 2, dev, seed=vdist.shard_seed(0, 0))
        train_step(wrapped, data)
        torch.cuda.synchronize()
        g1 = model.beta.grad.clone()
        model.zero_grad(set_to_none=True)
        train_step(model, data)                      # the same step without the wrapper
        close(g1.reshape(1), model.beta.grad.reshape(1), atol=1e-6, rtol=1e-5, what="GradSync(world 1) grad_beta")
        assert vdist.max_over_ranks(1.25, dev) == pytest.approx(1.25)
        vdist.barrier()
    finally:
        vdist.shutdown()


@pytest.mark.parametrize("amp_dtype", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
def test_backbone_under_autocast(dev, amp_dtype):
    """The drop-in module under the reference's training recipe (Lightning precision=16 = fp16
    autocast, base_cli.py:77, 90; and bf16): forward + backward run and every output of the path is
    fp32 and finite (the hot path computes in fp32 either way).  No closeness to the
    fp32 run is asserted beyond the mean: with random-init weights the half-precision convolutions
    in front of the Laplace density (beta = 0.1) move individual rays by O(1)."""
    from vampire_amd.backbone import BaseVAMPIRE2
    c = CFG_TINY
    kw = dict(x_bound_seg=list(c.x_bound_seg), y_bound_seg=list(c.y_bound_seg), z_bound_seg=list(c.z_bound_seg),
              x_bound_det=list(c.x_bound_det), y_bound_det=list(c.y_bound_det), z_bound_det=list(c.z_bound_det),
              d_bound=list(c.d_bound), final_dim=c.final_dim, downsample_factor=4, upsample_factor=4,
              mid_channels=4, output_channels=8, img_backbone_conf=dict(), img_neck_conf=dict(out_channels=[8] * 4),
              num_classes=5, density_mode="sdf", sdf_bias=-1.0, cat_pos=True, cat_seg=False)
    torch.manual_seed(0)
    mod = BaseVAMPIRE2(**kw).to(dev)
    with torch.no_grad():
        mod.density_conv.bias.fill_(-1.0)
    B = 2
    s2e, K, ida = synthetic.camera_rig(c, B, src_hw=(64, 176), focal=60.0, centre=(88.0, 34.0), jitter=2.0, seed=4)
    s2e[:, :, :3, 3] *= 0.3
    mats = dict(sensor2ego_mats=s2e[:, None], intrin_mats=K[:, None], ida_mats=ida[:, None],
                sensor2sensor_mats=torch.eye(4).expand(B, 1, 6, 4, 4), bda_mat=synthetic.bda_matrix(B, rot_deg=8.0))
    mats = {k: v.to(dev) for k, v in mats.items()}
    imgs = torch.randn(B, 1, 6, 3, *c.final_dim, device=dev)
    pts = [torch.rand(50, 3, device=dev) * 8 - 4 for _ in range(B)]
    ref = mod(imgs, mats, inrange_pts=pts)
    x = imgs.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=amp_dtype):
        out = mod(x, mats, inrange_pts=pts)
        loss = sum((o.float() ** 2).mean() for o in out[:8])
    loss.backward()
    assert bool(torch.isfinite(x.grad).all()) and float(x.grad.abs().max()) > 0
    assert bool(torch.isfinite(mod.density.beta.grad))
    for i in range(1, 8):
        assert out[i].dtype == torch.float32, i
        assert bool(torch.isfinite(out[i]).all()), i
        assert abs(float(out[i].mean()) - float(ref[i].mean())) <= 0.1 * float(ref[i].abs().mean()) + 0.05, i


@pytest.mark.parametrize("name", ["BaseLSSImpaintor", "BaseLSS", "BaseBiLinear"])
def test_sibling_backbones_forward(dev, name):
    """The drop-in mirrors of the other three backbones (base_lss_impaintor.py:79, base_lss.py:16,
    base_bilinear.py:80) end to end on the GPU against the same module with lift / render / point
    queries swapped for the oracle on the CPU: static occupancy grid (no bda rotation), Conv3d +
    Softplus base, and for BaseBiLinear the D = 1 lift and feature_conv."""
    import copy
    from oracle import aten_oracle as O
    import vampire_amd.backbone as BB
    c = CFG_TINY
    cls = getattr(BB, name)
    kw = dict(x_bound_seg=list(c.x_bound_seg), y_bound_seg=list(c.y_bound_seg), z_bound_seg=list(c.z_bound_seg),
              x_bound_det=list(c.x_bound_det), y_bound_det=list(c.y_bound_det), z_bound_det=list(c.z_bound_det),
              d_bound=list(c.d_bound), final_dim=c.final_dim, downsample_factor=4, upsample_factor=4,
              mid_channels=4, output_channels=8, img_backbone_conf=dict(), img_neck_conf=dict(out_channels=[8] * 4),
              num_classes=5, density_mode="sdf", sdf_bias=-1.0)
    torch.manual_seed(1)
    ref = cls(**kw).eval()
    with torch.no_grad():
        ref.density_conv.bias.fill_(-1.0)
    mod = copy.deepcopy(ref).to(dev)
    B = 2
    s2e, K, ida = synthetic.camera_rig(c, B, src_hw=(64, 176), focal=60.0, centre=(88.0, 34.0), jitter=2.0, seed=4)
    s2e[:, :, :3, 3] *= 0.3
    bda = synthetic.bda_matrix(B, rot_deg=8.0)
    mats = dict(sensor2ego_mats=s2e[:, None], intrin_mats=K[:, None], ida_mats=ida[:, None],
                sensor2sensor_mats=torch.eye(4).expand(B, 1, 6, 4, 4), bda_mat=bda)
    imgs = torch.randn(B, 1, 6, 3, *c.final_dim)
    out = mod(imgs.to(dev), {k: v.to(dev) for k, v in mats.items()})
    assert len(out) == 12
    with torch.no_grad():
        m = ref
        feats = m.get_cam_feats(imgs)
        src = feats[:, 0].reshape(B * 6, -1, feats.shape[-2], feats.shape[-1])
        feat = m.channel_lower(src).reshape(B, 6, -1, *src.shape[-2:])
        if m._USE_DEPTH:
            depth = m.mapping_along_depth(src).softmax(dim=1).reshape(B, 6, -1, *src.shape[-2:])
            vox = O.lift(depth, feat, m.voxel_coords, s2e, K, ida, bda, c.final_dim, c.d_bound)
        else:
            pix = O.ego_to_pixel(m.voxel_coords, s2e, K, ida, bda)
            vox = O.lift_from_frustum_feats(feat.unsqueeze(3), pix, c.final_dim, c.d_bound, use_depth=False)
        if m.cat_pos:
            vox = torch.cat([vox, m.norm_voxel_coords.permute(3, 0, 1, 2)[None].repeat(B, 1, 1, 1, 1)], 1)
        base = m.base_conv(vox)
        dens, sem = m.density_conv(base), m.seg_conv(base)
        if not m._USE_DEPTH:
            base = m.feature_conv(base)
        rgb = m.rgb_conv(base)
        geom = torch.nan_to_num(O.frustum_to_ego(m.frustum, s2e, K, ida, bda), -1e3)
        r = O.render(geom, dens, sem, base, rgb, seg_bounds=(c.x_bound_seg, c.y_bound_seg, c.z_bound_seg),
                     output_coords=m.output_coords, camera_mids=m.camera_mids, bev_mids=m.bev_mids,
                     d_far=c.d_bound[1], z_step_det=c.z_bound_det[2], num_classes=5, density_mode="sdf",
                     beta_param=m.density.beta, sdf_bias=-1.0, cat_seg=m.cat_seg)
        up = lambda t: m.upsample2d(t.reshape(B * 6, -1, m.fH, m.fW)).reshape(B, 6, -1, m.fH * 4, m.fW * 4)
        close(out[1], up(r[0]), atol=2e-4, what=name + " rgb_preds")
        close(out[2], up(r[1]), atol=2e-4, what=name + " seg_logits_preds")
        close(out[3], up(r[2]), atol=5e-4, what=name + " depth_preds")
        for i, j in ((4, 3), (5, 4), (6, 5), (7, 6)):
            close(out[i], r[j], atol=2e-4, what=f"{name} output {i}")
        bev_feat = m.voxel_output((r[7] * r[6].tanh()).reshape(B, -1, *r[7].shape[-2:]))
        close(out[0], bev_feat, atol=5e-4, what=name + " bev feature")
        # static occupancy grid: F.grid_sample on the module's own norm_occ_coords buffer
        import torch.nn.functional as F
        g = m.norm_occ_coords[None].expand(B, *m.norm_occ_coords.shape)
        close(out[10], F.grid_sample(sem, g, padding_mode="border", align_corners=True).permute(0, 2, 3, 4, 1),
              atol=2e-4, what=name + " occ_logits")
        close(out[11], F.grid_sample(m.density(dens), g, align_corners=True).permute(0, 2, 3, 4, 1).tanh(),
              atol=2e-4, what=name + " occ_density")


@pytest.mark.parametrize("name", ["BaseLSSImpaintor", "BaseLSS", "BaseBiLinear"])
def test_sibling_backbones_backward(dev, name):
    """Gradients through the sibling backbones on the GPU (HIP operators) against the same module with the
    oracle operators on CPU: images, a few parameters, beta -- the D = 1 lift's backward and the static
    occupancy grid included (these were forward-only)."""
    import copy
    import vampire_amd.backbone as BB
    from tests.test_multitask import OracleOps
    c = dataclasses.replace(CFG_TINY, density_mode="sdf", cat_seg=True)
    cls = getattr(BB, name)
    kw = dict(x_bound_seg=list(c.x_bound_seg), y_bound_seg=list(c.y_bound_seg), z_bound_seg=list(c.z_bound_seg),
              x_bound_det=list(c.x_bound_det), y_bound_det=list(c.y_bound_det), z_bound_det=list(c.z_bound_det),
              d_bound=list(c.d_bound), final_dim=c.final_dim, downsample_factor=4, upsample_factor=4,
              mid_channels=4, output_channels=8, img_backbone_conf=dict(), img_neck_conf=dict(out_channels=[8] * 4),
              num_classes=5, density_mode="sdf", sdf_bias=-1.0)
    torch.manual_seed(1)
    ref = cls(**kw).eval()
    with torch.no_grad():
        ref.density_conv.bias.fill_(-1.0)
    mod = copy.deepcopy(ref).to(dev)
    ref._hot = OracleOps(dataclasses.replace(c, cat_seg=ref.cat_seg))
    B = 2
    s2e, K, ida = synthetic.camera_rig(c, B, src_hw=(64, 176), focal=60.0, centre=(88.0, 34.0), jitter=2.0, seed=4)
    s2e[:, :, :3, 3] *= 0.3
    mats = dict(sensor2ego_mats=s2e[:, None], intrin_mats=K[:, None], ida_mats=ida[:, None],
                sensor2sensor_mats=torch.eye(4).expand(B, 1, 6, 4, 4), bda_mat=synthetic.bda_matrix(B, rot_deg=8.0))
    imgs = torch.randn(B, 1, 6, 3, *c.final_dim)
    pts = [torch.rand(40, 3) * 8 - 4 for _ in range(B)]
    a, b_ = imgs.clone().to(dev).requires_grad_(True), imgs.clone().requires_grad_(True)
    out_d = mod(a, {k: v.to(dev) for k, v in mats.items()}, inrange_pts=[p.to(dev) for p in pts])
    out_r = ref(b_, mats, inrange_pts=pts)
    tens = lambda o: [t for t in o[:8]] + list(o[8]) + list(o[9]) + [o[10], o[11]]
    for i, (x, y) in enumerate(zip(tens(out_d), tens(out_r))):
        close(x, y, atol=5e-4, rtol=1e-3, what=f"{name} output {i}")
    loss = lambda o: sum((t.float() ** 2).mean() for t in tens(o))
    loss(out_d).backward(); loss(out_r).backward()
    close(a.grad, b_.grad, atol=1e-6, rtol=5e-3, scale="max", what=name + " grad images")
    pr, pd = dict(ref.named_parameters()), dict(mod.named_parameters())
    names = ["channel_lower.weight", "density_conv.weight", "seg_conv.bias", "rgb_conv.0.weight", "density.beta"]
    names += ["mapping_along_depth.0.weight"] if ref._USE_DEPTH else ["feature_conv.weight"]
    for n in names:
        close(pd[n].grad, pr[n].grad, atol=1e-6, rtol=5e-3, scale="max", what=f"{name} grad {n}")


# --------------------------------------------------------------------------- round-3 parity pins: early ray termination
def _regime_inputs(cfg, regime, dev, with_grad=True):
    """The render inputs of tests/golden/make_golden.py: make_regimes ("sdf" = the bench workload)."""
    mode = "naive" if regime == "naive" else "sdf"
    cfg = dataclasses.replace(cfg, density_mode=mode)
    vols = list(synthetic.render_inputs(cfg, 1, seed=0, device=dev))
    # init: sdf_bias - 10 = density_conv's initial bias (bv2:241), sigma = 1 / beta everywhere;
    # empty: s - bias ~ +0.6, sigma ~ 0.012 / m, no ray ever saturates
    scale, shift = {"init": (1.0, -10.0), "empty": (0.4, 0.0)}.get(regime, (1.0, 0.0))
    vols[0] = vols[0] * scale + shift
    if with_grad:
        vols = [v.detach().requires_grad_(True) for v in vols]
    return cfg, vols


def _render_fwd_bwd(cfg, vols, rm, dev, ert, seed, beta_value=0.1, cam_direct=None):
    hp = hot(cfg, dev)
    hp.impl["ert"] = ert
    if cam_direct is not None:          # (default "auto": one kernel with early termination, copy + march without)
        hp.impl["cam_direct"] = cam_direct
    for v in vols:
        v.grad = None
    beta = torch.tensor(beta_value, device=dev, requires_grad=(cfg.density_mode == "sdf"))
    outs = hp.render(*vols, beta if cfg.density_mode == "sdf" else None, render_mats=rm)
    torch.autograd.backward(outs, _upstream([o.shape for o in outs], seed, dev))
    grads = [v.grad.clone() for v in vols]
    return [o.detach().clone() for o in outs], grads, (beta.grad.clone() if beta.grad is not None else None)


@pytest.mark.parametrize("direct", [True, False], ids=["one-kernel", "copy+march"])
@pytest.mark.parametrize("regime", ["sdf", "naive", "init", "empty"])
def test_ert_on_equals_off_full_size(dev, regime, direct):
    """Early ray termination (render_common.hpp) against the same kernels with it switched off (both camera
    forwards: by default the one-kernel forward runs with termination, copy + march without), on
    identical cfg-B inputs, in the three density regimes: the synthetic sdf workload (64 % of the
    inside samples dropped), density_mode="naive" (rays saturate behind the volume: the analytic
    exit of cam_term_kernel), the reference's initial regime (sigma = 1 / beta: every ray saturates
    within three samples) and an empty scene (nothing terminates).  All 8
    outputs within 1e-6 and all gradients within 1e-5 of the tensor's largest magnitude."""
    with open(os.path.join(GOLDEN, "full_checksums.json")) as f:
        rm = torch.tensor(json.load(f)["B"]["render_mats"], dtype=torch.float32, device=dev)
    cfg, vols = _regime_inputs(CFG_B, regime, dev)
    on = _render_fwd_bwd(cfg, vols, rm, dev, True, 4545, cam_direct=direct)
    off = _render_fwd_bwd(cfg, vols, rm, dev, False, 4545, cam_direct=direct)
    for nm, a, b in zip(NAMES, on[0], off[0]):
        # (relative to the tensor's largest magnitude: depth_preds = sum w mid + (1 - sum w) * 70.4 carries
        # the fp32 rounding of sum w ~ 1 times d_far, about 1e-5 absolute, whatever the order of summation)
        # three fp32 roundings of sum w ~ 1, times d_far = 70.4, for the depth map)
        close(a, b, atol=(3 * 1.2e-7 * cfg.d_bound[1] if nm == "depth_preds" else 1e-7), rtol=1e-6, scale="max",
              what=f"{regime} ERT on/off {nm}")
    for nm, a, b in zip(("density_feature", "semantic_logits", "base", "rgb"), on[1], off[1]):
        close(a, b, atol=1e-12, rtol=1e-5, scale="max", what=f"{regime} ERT on/off grad_{nm}")
    if on[2] is not None:
        assert abs(float(on[2]) - float(off[2])) <= 1e-5 * abs(float(off[2])) + 1e-9, (float(on[2]), float(off[2]))


@pytest.mark.parametrize("regime", ["sdf", "naive", "init", "empty"])
def test_camera_forward_is_chosen_from_the_data(dev, regime):
    """cam_direct = "auto" (the default): the camera forward of a call follows what the rays of EARLIER calls did
    (HotPath._camera_forward_choice: the termination table's statistic, probed at fixed call indices and applied at
    a fixed later call index behind an event -- deterministic: the sequence of modes is asserted call by call, with no
    synchronisation by the test).  Where rays saturate (sdf workload, the reference's initial regime) it stays with
    the one kernel + early termination; where they do not (sigmoid density, empty scene) it goes to copy + planned
    march at the call that applies the first probe -- and whichever path a call takes, outputs and gradients are those
    of the termination-off path (bv2:191-194: density_mode='naive' is a first-class mode of the reference)."""
    with open(os.path.join(GOLDEN, "full_checksums.json")) as f:
        rm = torch.tensor(json.load(f)["B"]["render_mats"], dtype=torch.float32, device=dev)
    cfg, vols = _regime_inputs(CFG_B, regime, dev)
    # (the two forwards agree to the rounding of their sample coordinates, 3e-5 of a map's scale; a call is held
    # to the termination-off result of the forward it took, at the tolerance of test_ert_on_equals_off_full_size)
    refs = {"planned": _render_fwd_bwd(cfg, vols, rm, dev, False, 4545, cam_direct=False),
            "direct": _render_fwd_bwd(cfg, vols, rm, dev, False, 4545, cam_direct=True)}
    hp = hot(cfg, dev)
    assert hp.impl["cam_direct"] == "auto" and hp.impl["ert"]
    hp._PROBE_EVERY = 1
    seen = []
    for it in range(4):
        for v in vols:
            v.grad = None
        beta = torch.tensor(0.1, device=dev, requires_grad=(cfg.density_mode == "sdf"))
        outs = hp.render(*vols, beta if cfg.density_mode == "sdf" else None, render_mats=rm)
        torch.autograd.backward(outs, _upstream([o.shape for o in outs], 4545, dev))
        seen.append(hp.camera_forward_mode())               # (the mode this call ran in: it is decided as a call starts)
        off = refs[seen[-1]]
        for nm, a, b in zip(NAMES, outs, off[0]):
            close(a.detach(), b, atol=(3 * 1.2e-7 * cfg.d_bound[1] if nm == "depth_preds" else 1e-7), rtol=1e-6, scale="max",
                  what=f"{regime} call {it} ({seen}) {nm}")
        for nm, v, b in zip(("density_feature", "semantic_logits", "base", "rgb"), vols, off[1]):
            close(v.grad, b, atol=1e-12, rtol=1e-5, scale="max", what=f"{regime} call {it} ({seen}) grad_{nm}")
    r = hp._cam_sel["last"]
    want = "direct" if regime in ("sdf", "init") else "planned"
    # probe every call, applied one call later: call 0 runs the one kernel, every later call what the data asks for
    assert seen == ["direct"] + [want] * 3, (regime, seen, r)
    assert (r < 0.4) if want == "direct" else (r > 0.5), (regime, r)
    # the default cadence: probe at calls 1, 17, ..., applied 8 calls on -- the switch happens AT call 9, in every run
    hp2 = hot(cfg, dev)
    modes = []
    with torch.no_grad():
        for it in range(12):
            hp2.render(*[v.detach() for v in vols], torch.tensor(0.1, device=dev) if cfg.density_mode == "sdf" else None,
                       render_mats=rm)
            modes.append(hp2.camera_forward_mode())
    assert modes == ["direct"] * 8 + [want] * 4, (regime, modes)


@pytest.mark.parametrize("ert", [True, False], ids=["ert", "no-ert"])
@pytest.mark.parametrize("regime", ["naive", "init", "empty"])
def test_density_regimes_match_reference(dev, regime, ert):
    """cfg-B in the regimes of tests/golden/regime_checksums.json (the REFERENCE run with autograd):
    sigmoid density -- where the analytic exit of the termination table fires --, the reference's
    initial density regime (everything saturates) and an empty scene (nothing does), with early ray
    termination on and off: block statistics of
    the eight outputs, of the four volume gradients and grad_beta."""
    with open(os.path.join(GOLDEN, "regime_checksums.json")) as f:
        ref = json.load(f)[regime]
    with open(os.path.join(GOLDEN, "full_checksums.json")) as f:
        rm = torch.tensor(json.load(f)["B"]["render_mats"], dtype=torch.float32, device=dev)
    cfg, vols = _regime_inputs(CFG_B, regime, dev)
    assert cfg.density_mode == ref["density_mode"]
    outs, grads, gbeta = _render_fwd_bwd(cfg, vols, rm, dev, ert, ref["seed_render"],
                                         beta_value=ref.get("beta", 0.1))
    # elem_atol: where sigma = (1 + expm1(-|t| / beta)) / (2 beta) has expm1 within a few ulp of -1 it is
    # the last bits of expm1f (CPU and GPU libm differ there); an absolute floor of 1e-8 per element,
    # four orders under the 1e-4 bar, keeps that out of the check
    for nm, o in zip(NAMES, outs):
        _block_check(o, ref[nm], f"{regime} {nm}", rtol=5e-5, elem_atol=1e-8)
    for k, g in zip(("density_feature", "semantic_logits", "base", "rgb"), grads):
        _block_check(g, ref["grad_" + k], f"{regime} grad_{k}", rtol=5e-5, elem_atol=1e-10)
    if "grad_beta" in ref:
        assert abs(float(gbeta) - ref["grad_beta"]) <= 2e-3 * abs(ref["grad_beta"]) + 1e-4, (float(gbeta), ref["grad_beta"])


def test_debug_checks_catch_broken_promises(dev):
    """vamp_debug_checks(1): the *_CLEAN / *_VALID promises of the host are verified before the library relies on
    them.  A training step of the default schedule passes every check; a workspace whose counters are not zero, or
    that holds no termination table, is refused with "promise broken" instead of corrupting the result."""
    from vampire_amd import _capi
    from vampire_amd.step import LiftRenderStep, SyntheticBatch, train_step
    cfg = dataclasses.replace(CFG_TINY, density_mode="sdf")
    lib = _capi.load()
    lib.vamp_debug_checks(1)
    try:
        model = LiftRenderStep(cfg, dev)
        data = SyntheticBatch(cfg, 2, dev)
        for _ in range(3):                     # (CELLS_CLEAN / COUNTERS_CLEAN / TERM_VALID all in use from the second step on)
            model.zero_grad(set_to_none=True)
            train_step(model, data)
        torch.cuda.synchronize()
        hp = model.hp
        # counters not zero under VAMP_LIFTFWD_CELLS_CLEAN
        hp._ws["lift"].fill_(1)
        with pytest.raises(_capi.VampireHipError, match="promise broken.*CELLS_CLEAN"):
            train_step(model, data)
        hp._ws["lift"].zero_(); hp._dirty.discard("lift")
        # counters not zero under VAMP_CAMPREP_COUNTERS_CLEAN (the forward's prepare pass)
        hp._ws["render"].fill_(1); hp._dirty.discard("render")
        with pytest.raises(_capi.VampireHipError, match="promise broken"):
            train_step(model, data)
        hp._ws["render"].zero_(); hp._dirty.discard("render")
        # a termination table that is not one, handed to the planned march with VAMP_CAMFWD_TERM_VALID
        hp.impl["cam_direct"] = False
        d = hp.render_desc(2, cfg.num_cams, _capi.VAMP_F32)
        off = hp.lib.vamp_render_term_offset(C.byref(d))
        ws = hp._workspace("render", hp.lib.vamp_render_workspace_bytes(C.byref(d)))
        ws[off:off + 4 * 2 * cfg.num_cams * cfg.fH * cfg.fW].view(torch.int32).fill_(-7)
        rgb_p = torch.empty(2, cfg.num_cams, 3, cfg.fH, cfg.fW, device=dev)
        seg_p = torch.empty(2, cfg.num_cams, cfg.num_classes, cfg.fH, cfg.fW, device=dev)
        dep_p = torch.empty(2, cfg.num_cams, 1, cfg.fH, cfg.fW, device=dev)
        ptr = lambda t: C.c_void_p(t.data_ptr())
        vols = [v.detach() for v in data.vols]
        beta = torch.tensor([0.1], device=dev)
        rc = hp.lib.vamp_render_camera_forward_ex(
            C.byref(d), None, ptr(data.render_mats), ptr(hp.us), ptr(hp.vs), ptr(hp.ds), ptr(hp.camera_mids), ptr(beta),
            ptr(vols[0]), ptr(vols[1]), ptr(vols[3]), ptr(rgb_p), ptr(seg_p), ptr(dep_p), ptr(ws), ws.numel(),
            _capi.VAMP_CAMFWD_TERM_VALID, None)
        assert rc == -1 and b"promise broken" in hp.lib.vamp_last_error() and b"TERM_VALID" in hp.lib.vamp_last_error()
    finally:
        lib.vamp_debug_checks(0)
        torch.cuda.synchronize()


# --------------------------------------------------------------------------- round-3: one-kernel camera forward
@pytest.mark.parametrize("mode,cat_seg", RENDER_VARIANTS)
def test_camera_direct_forward_tiny(tiny_common, dev, mode, cat_seg):
    """render_cam_direct.hip (the camera branch as one kernel on the channel-first volumes; what
    forward-only calls run) against the reference's outputs on the tiny fixtures, with and without
    early ray termination, fp32 and bf16-rounded inputs; and against the packed-copy path bit for
    bit in the termination table."""
    g = tiny_common
    r = load_golden(render_fixture_name(mode, cat_seg))
    cfg = dataclasses.replace(CFG_TINY, density_mode=mode, cat_seg=cat_seg)
    _, rm = tiny_mats(g, dev)
    vols = [g[k].to(dev) for k in ("density_feature", "semantic_logits", "base", "rgb")]
    beta = r["beta"].reshape(()).to(dev) if mode == "sdf" else None
    for ert in (True, False):
        hp = hot(cfg, dev)
        hp.impl["ert"] = ert
        hp.impl["cam_direct"] = True          # (the default; explicit so that VAMP_CAM_DIRECT=0 runs of the suite still test this kernel)
        with torch.no_grad():
            outs = hp.render(*vols, beta, render_mats=rm)
        for nm, o in zip(NAMES, outs):
            close(o, r[nm], what=f"direct {mode} ert={ert} {nm}")
        hp.impl["cam_direct"] = False
        with torch.no_grad():
            old = hp.render(*vols, beta, render_mats=rm)
        for nm, a, b_ in zip(NAMES[:3], outs, old):
            close(a, b_, atol=5e-5, rtol=2e-5, what=f"direct vs packed {mode} ert={ert} {nm}")
    # bf16 volumes: the kernel reads the 2-byte elements itself
    hp = hot(cfg, dev)
    vb = [v.bfloat16() for v in vols]
    with torch.no_grad():
        ob = hp.render(*vb, beta, render_mats=rm)
        of = hp.render(*[v.float() for v in vb], beta, render_mats=rm)
    for nm, a, b_ in zip(NAMES, ob, of):
        close(a, b_, atol=1e-5, rtol=1e-5, what=f"direct bf16 {nm}")


@pytest.mark.parametrize("regime", ["sdf", "naive", "init", "empty"])
def test_camera_direct_forward_full_size(dev, regime):
    """cfg-B: the one-kernel camera forward against the packed-copy march in the four density
    regimes (outputs to 1e-5 of the largest magnitude) and, where the reference fixture exists,
    against the reference's block statistics; its termination table equals cam_term_kernel's."""
    with open(os.path.join(GOLDEN, "full_checksums.json")) as f:
        rm = torch.tensor(json.load(f)["B"]["render_mats"], dtype=torch.float32, device=dev)
    cfg, vols = _regime_inputs(CFG_B, regime, dev, with_grad=False)
    beta = torch.tensor(0.1, device=dev) if cfg.density_mode == "sdf" else None
    hp = hot(cfg, dev)
    with torch.no_grad():
        new = hp.render(*vols, beta, render_mats=rm)
        hp.impl["cam_direct"] = False
        old = hp.render(*vols, beta, render_mats=rm)
    # (the one-kernel forward evaluates the sample positions from a per-ray line set up in fp64 -- the
    # correctly rounded tap coordinates -- where the packed march repeats the reference's fp32 chain bit
    # for bit: the two differ by the reference's own rounding, ~1e-5 voxel, which on these white-noise
    # volumes, the worst case, moves a composited value by up to 3e-5; the bar is 1e-4)
    for nm, a, b_ in zip(NAMES[:3], new, old):
        close(a, b_, atol=(3 * 1.2e-7 * cfg.d_bound[1] if nm == "depth_preds" else 1e-7), rtol=5e-5, scale="max",
              what=f"{regime} direct vs packed {nm}")
    if regime != "sdf":
        with open(os.path.join(GOLDEN, "regime_checksums.json")) as f:
            ref = json.load(f)[regime]
        for nm, o in zip(NAMES, new):
            _block_check(o, ref[nm], f"{regime} direct {nm}", rtol=5e-5, elem_atol=1e-8)


# --------------------------------------------------------------------------- round 6: the render forward as one launch
@pytest.mark.parametrize("mode,cat_seg", RENDER_VARIANTS)
def test_render_forward_merged_tiny(tiny_common, dev, mode, cat_seg):
    """render_fwd_merged.hip (camera tiles + BEV column blocks in one grid; what every forward with early
    termination runs) against the reference's eight outputs on the tiny fixtures and BIT FOR BIT against the two
    launches it replaces -- forward-only and in a training call (saved samples: the gradients follow)."""
    g = tiny_common
    r = load_golden(render_fixture_name(mode, cat_seg))
    cfg = dataclasses.replace(CFG_TINY, density_mode=mode, cat_seg=cat_seg)
    _, rm = tiny_mats(g, dev)
    names = ("density_feature", "semantic_logits", "base", "rgb")
    res = {}
    for merged in (True, False):
        hp = hot(cfg, dev)
        hp.impl.update(cam_direct=True, ert=True, fwd_merged=merged)
        vols = [g[k].to(dev) for k in names]
        beta = r["beta"].reshape(()).to(dev) if mode == "sdf" else None
        with torch.no_grad():
            nog = hp.render(*vols, beta, render_mats=rm)
        vols = [v.clone().requires_grad_(True) for v in vols]
        if beta is not None:
            beta = beta.clone().requires_grad_(True)
        outs = hp.render(*vols, beta, render_mats=rm)
        torch.autograd.backward(outs, [r["g_" + n].to(dev) for n in NAMES])
        res[merged] = (nog, outs, [v.grad for v in vols] + ([beta.grad] if beta is not None else []))
        for nm, o in zip(NAMES, outs):
            close(o, r[nm], what=f"merged={merged} {mode} {nm}")
        for k, v in zip(names, vols):
            close(v.grad, r["grad_" + k], atol=1e-5, rtol=1e-4, scale="max", what=f"merged={merged} grad_{k}")
    for part, what in enumerate(("no-grad outputs", "training outputs")):
        for nm, a, b_ in zip(NAMES, res[True][part], res[False][part]):
            assert torch.equal(a, b_), f"{what}: merged launch differs from the two launches in {nm}"
    for a, b_ in zip(res[True][2], res[False][2]):
        close(a, b_, atol=1e-6, rtol=1e-6, scale="max", what="gradients behind the merged forward")


@pytest.mark.parametrize("name,cfg,batch", [("B", CFG_B, 1), ("B", CFG_B, 3), ("A", CFG_A, 1)])
def test_render_forward_merged_full_size(dev, name, cfg, batch):
    """cfg-A / cfg-B (and a batch: the BEV blocks' (sample, channel group) decode): the merged launch's eight
    outputs and its termination table equal the two launches' bit for bit."""
    from vampire_amd.step import SyntheticBatch
    data = SyntheticBatch(cfg, batch, dev, seed=11)
    beta = torch.tensor(0.1, device=dev)
    hp = hot(cfg, dev)
    hp.impl.update(cam_direct=True, ert=True)
    got = {}
    for merged in (True, False):
        hp.impl["fwd_merged"] = merged
        with torch.no_grad():
            outs = hp.render(*[v.detach() for v in data.vols], beta, render_mats=data.render_mats)
        d = hp.render_desc(batch, cfg.num_cams, _capi.VAMP_F32)
        off = hp.lib.vamp_render_term_offset(C.byref(d))
        n = batch * cfg.num_cams * cfg.fH * cfg.fW
        term = hp._ws["render"][off:off + 4 * n].view(torch.int32).clone()
        got[merged] = (outs, term)
    for nm, a, b_ in zip(NAMES, got[True][0], got[False][0]):
        assert torch.equal(a, b_), f"cfg-{name} x{batch}: merged launch differs from the two launches in {nm}"
    assert torch.equal(got[True][1], got[False][1]), "termination table"


def test_bev_forward_checks_the_heights(tiny_common, dev):
    """The one-kernel BEV forward sizes its plane slabs for the reference's lattice of heights and would clamp a
    plane outside them; the library therefore checks the caller's host copy of `ozs` and sends anything else -- here
    heights spread over the whole volume -- to the two-kernel path (advisor finding of round 3): same results as
    the two-kernel path asked for explicitly, and the merged launch refuses such heights."""
    g = tiny_common
    cfg = dataclasses.replace(CFG_TINY, density_mode="sdf", cat_seg=False)
    _, rm = tiny_mats(g, dev)
    vols = [g[k].to(dev) for k in ("density_feature", "semantic_logits", "base", "rgb")]
    beta = torch.tensor(0.1, device=dev)
    hp = hot(cfg, dev)
    hp.impl["cam_direct"] = True
    d = hp.render_desc(2, cfg.num_cams, _capi.VAMP_F32)
    assert hp.lib.vamp_render_forward_merged_supported(C.byref(d), hp.ozs_host) == 1
    assert hp.lib.vamp_render_forward_merged_supported(C.byref(d), None) == 0
    # heights spread over the whole z range of the volume, unevenly: far more planes than a lattice with the det
    # grid's spacing touches
    lo, hi = cfg.z_bound_seg[0], cfg.z_bound_seg[1]
    wild = torch.linspace(lo + 0.05, hi - 0.05, cfg.oZ) + 0.07 * torch.sin(torch.arange(cfg.oZ).float())
    wild = wild.clamp(lo + 0.01, hi - 0.01)
    hp.ozs = wild.to(dev)
    hp.ozs_host = (C.c_float * cfg.oZ)(*wild.tolist())
    assert hp.lib.vamp_render_forward_merged_supported(C.byref(d), hp.ozs_host) == 0
    with torch.no_grad():
        auto = hp.render(*vols, beta, render_mats=rm)              # fused asked for, heights do not fit
        hp.impl["bev_fused"] = False
        two = hp.render(*vols, beta, render_mats=rm)
    for nm, a, b_ in zip(NAMES, auto, two):
        assert torch.equal(a, b_), f"non-lattice heights: {nm} differs from the two-kernel path"
    # and the values are those of aten's grid_sample at these heights (the oracle with the same det lattice in x / y)
    from oracle import aten_oracle as O
    geo = PathGeometry(cfg)
    oc = geo.output_coords.clone()
    oc[..., 2] = wild.reshape(-1, 1, 1)
    geom = torch.nan_to_num(g["geom"], -1e3)
    ref = O.render(geom, *[v.cpu() for v in vols],
                   seg_bounds=(cfg.x_bound_seg, cfg.y_bound_seg, cfg.z_bound_seg), output_coords=oc,
                   camera_mids=geo.camera_mids, bev_mids=geo.bev_mids, d_far=cfg.d_bound[1],
                   z_step_det=cfg.z_bound_det[2], num_classes=cfg.num_classes, density_mode=cfg.density_mode,
                   beta_param=torch.tensor(0.1), sdf_bias=cfg.sdf_bias, cat_seg=cfg.cat_seg)
    for nm, a, b_ in zip(NAMES[3:], auto[3:], ref[3:]):
        close(a, b_, what=f"non-lattice heights vs oracle {nm}")


# --------------------------------------------------------------------------- N4: configs[4] end to end
def test_multitask_step_matches_oracle_operators(dev):
    """The full multi-task model (R18 encoder, backbone, BEV head, nine losses) with the HIP operators on the
    GPU against the same weights and batch with the oracle operators on CPU: every loss term."""
    import copy
    import dataclasses
    from vampire_amd import multitask as M
    from tests.test_multitask import OracleOps
    cfg = dataclasses.replace(CFG_TINY, density_mode="sdf", final_dim=(192, 224), num_classes=6)
    torch.manual_seed(0)
    bb, hd = M.reference_confs(cfg, output_channels=8, small_encoder=True)
    ref = M.VAMPIRE2(bb, hd)
    with torch.no_grad():
        ref.backbone.density_conv.bias.fill_(cfg.sdf_bias)
    mod = copy.deepcopy(ref).to(dev)
    ref.backbone._hot = OracleOps(cfg)
    batch = M.synthetic_batch(cfg, 2, seed=5, num_points=40, num_boxes=6)
    batch_d = M.synthetic_batch(cfg, 2, seed=5, num_points=40, num_boxes=6, device=dev)
    lf_r, lf_d = M.MultiTaskLoss(ref, sdf_bias=cfg.sdf_bias), M.MultiTaskLoss(mod, sdf_bias=cfg.sdf_bias)
    loss_r = M.multitask_step(ref, lf_r, batch, amp_dtype=None)
    loss_d = M.multitask_step(mod, lf_d, batch_d, amp_dtype=None)
    for k in lf_r.last:
        a, b = float(lf_d.last[k]), float(lf_r.last[k])
        assert abs(a - b) <= 2e-3 * abs(b) + 1e-4, (k, a, b)
    assert abs(float(loss_d) - float(loss_r)) <= 2e-3 * abs(float(loss_r))
    # gradients of a few representative parameters (encoder stem, depth head, 3-D stem, BEV head)
    pr, pd = dict(ref.named_parameters()), dict(mod.named_parameters())
    for name in ("backbone.img_backbone.conv1.weight", "backbone.mapping_along_depth.0.weight",
                 "backbone.base_conv.init_dres.weight", "backbone.voxel_output.weight", "head.shared_conv.0.weight"):
        close(pd[name].grad, pr[name].grad, atol=1e-6, rtol=2e-2, scale="max", what="grad " + name)


def test_multitask_step_reference_config_bf16(dev):
    """configs[4] at the reference's own configuration (cfg-A, ResNet-50, 256 x 704, bf16 autocast), one
    sample: finite losses, a gradient on every parameter, stride-8 neck features feeding the lift."""
    from vampire_amd import multitask as M
    from vampire_amd.config import CFG_A
    torch.manual_seed(0)
    bb, hd = M.reference_confs(CFG_A)
    model = M.VAMPIRE2(bb, hd).to(dev)
    with torch.no_grad():
        model.backbone.density_conv.bias.fill_(CFG_A.sdf_bias)
    feats = model.backbone.get_cam_feats(torch.zeros(1, 1, 6, 3, 256, 704, device=dev))
    assert feats.shape == (1, 1, 6, 512, 32, 88)
    batch = M.synthetic_batch(CFG_A, 1, seed=2, device=dev, num_points=5000, num_boxes=20)
    lf = M.MultiTaskLoss(model, sdf_bias=CFG_A.sdf_bias)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4)
    loss = M.multitask_step(model, lf, batch, optimizer=opt)
    assert torch.isfinite(loss)
    assert all(torch.isfinite(torch.as_tensor(v)) for v in lf.last.values())
    assert not [n for n, p in model.named_parameters() if p.grad is None or not torch.isfinite(p.grad).all()]


# --------------------------------------------------------------------------- a11: BEVDepth-style voxel pooling
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("shape", [(2, 6, 7, 4, 5, 9, (16, 12, 1)), (1, 6, 112, 16, 44, 80, (128, 128, 1))],
                         ids=["tiny", "bevdepth"])
def test_voxel_pooling_against_numpy_definition(dev, shape, dtype):
    """north_star's pooling op (absent from the reference: parity UNPINNED) against a numpy scatter-add of the
    published definition: forward sums, backward rows; points outside the grid (incl. z) are dropped."""
    from oracle import voxel_pooling_oracle as VO
    from vampire_amd.ops import voxel_pooling
    B, N, D, H, W, C_, vn = shape
    gen = torch.Generator().manual_seed(11)
    geom = torch.stack([torch.randint(-3, vn[0] + 3, (B, N, D, H, W), generator=gen),
                        torch.randint(-3, vn[1] + 3, (B, N, D, H, W), generator=gen),
                        torch.randint(-1, vn[2] + 1, (B, N, D, H, W), generator=gen)], -1)
    geom[0, 0, 0] = torch.tensor([2, 3, 0])                       # one crowded cell
    feat = torch.randn(B, N, D, H, W, C_, generator=gen).to(dtype)
    f_d = feat.to(dev).requires_grad_(True)
    out = voxel_pooling(geom.to(dev), f_d, vn)
    assert out.shape == (B, C_, vn[1], vn[0])
    want = VO.voxel_pooling(geom.reshape(B, -1, 3).numpy(), feat.float().reshape(B, -1, C_).numpy(), vn)
    close(out, torch.from_numpy(want).float(), atol=1e-5, rtol=1e-5, scale="max", what="voxel_pooling")
    go = torch.randn(out.shape, generator=gen)
    out.backward(go.to(dev))
    gw = VO.voxel_pooling_backward(geom.reshape(B, -1, 3).numpy(), go.numpy(), vn)
    assert f_d.grad.dtype == dtype
    got = f_d.grad.float().cpu().reshape(B, -1, C_)
    ref = torch.from_numpy(gw).to(dtype).float()
    assert torch.equal(got, ref)
    # every point outside: an all-zero output (no zero fill needed by the caller)
    far = torch.full_like(geom, -5).to(dev)
    assert float(voxel_pooling(far, f_d.detach(), vn).abs().max()) == 0.0


# --------------------------------------------------------------------------- N3 in bf16 (the reference's precision=16)
@pytest.mark.parametrize("cin,cout,vol,batch", [(16, 16, (6, 12, 40), 2), (16, 32, (4, 8, 72), 1), (32, 16, (5, 8, 24), 1),
                                                (32, 32, (3, 4, 136), 1), (32, 32, (8, 100, 100), 1), (16, 16, (2, 4, 12), 1), (19, 16, (4, 7, 40), 2), (16, 22, (3, 5, 24), 1), (5, 7, (2, 4, 8), 1), (16, 16, (16, 200, 200), 1)])
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_conv3d_bf16_matches_torch(dev, cin, cout, vol, batch, dt):
    """bf16 3x3x3 conv on the bf16 matrix cores (fp32 accumulate) against torch's conv3d evaluated in fp64 on the
    same bf16-rounded operands: forward (bf16 output rounding), data gradient, weight gradient."""
    import torch.nn.functional as F
    from vampire_amd.ops import conv3d_bf16, conv3d_bf16_supported
    gen = torch.Generator(device=dev).manual_seed(4)
    x = torch.randn(batch, cin, *vol, device=dev, generator=gen).to(dt)
    w = (torch.randn(cout, cin, 3, 3, 3, device=dev, generator=gen) * 0.05).to(dt)
    go = torch.randn(batch, cout, *vol, device=dev, generator=gen).to(dt)
    assert conv3d_bf16_supported(x, w, (1, 1, 1), (1, 1, 1), None)
    a, wa = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y = conv3d_bf16(a, wa)
    assert y.dtype == dt
    y.backward(go)
    big = vol[0] * vol[1] * vol[2] > 100000
    rd = torch.float32 if big else torch.float64           # (fp64 conv3d on 640 k voxels is slow; fp32 reference there)
    b, wb = x.to(rd).requires_grad_(True), w.to(rd).requires_grad_(True)
    yr = F.conv3d(b, wb, padding=1)
    yr.backward(go.to(rd))
    # one bf16 rounding of the output (2^-9 relative) on top of fp32 accumulation
    close(y, yr, atol=1e-3, rtol=2.0 ** -8, what="conv3d_bf16 forward")
    close(a.grad, b.grad, atol=1e-3, rtol=2.0 ** -8, what="conv3d_bf16 grad_in")
    # the weight gradient is accumulated in fp32 and rounded once to bf16
    close(wa.grad, wb.grad, atol=1e-6, rtol=2.0 ** -7, scale="max", what="conv3d_bf16 grad_w")


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_resize_16bit(dev, dt):
    """The UNet's trilinear resize with 16-bit tensors (mixed precision): same taps and fp32 arithmetic as the
    fp32 kernel, one rounding at the output; forward and the gather backward."""
    import torch.nn.functional as F
    from vampire_amd.ops import upsample_trilinear
    gen = torch.Generator(device=dev).manual_seed(2)
    x = torch.randn(2, 5, 4, 9, 13, device=dev, generator=gen).to(dt)
    size = (8, 17, 26)
    a = x.clone().requires_grad_(True)
    y = upsample_trilinear(a, size)
    assert y.dtype == dt and y.shape == (2, 5) + size
    b = x.double().requires_grad_(True)
    yr = F.interpolate(b, size, mode="trilinear", align_corners=True)
    rel = 2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -10
    close(y, yr, atol=1e-3, rtol=rel, what="resize 16-bit forward")
    go = torch.randn(y.shape, device=dev, generator=gen).to(dt)
    y.backward(go); yr.backward(go.double())
    assert a.grad.dtype == dt
    close(a.grad, b.grad, atol=1e-2, rtol=4 * rel, what="resize 16-bit backward")


def test_round3_entry_points_reject_bad_arguments(dev):
    """Error behaviour of the round-3 entry points: a bad descriptor / dtype / workspace comes back as a negative
    code with a message (no launch, no crash); the Python host raises VampireHipError."""
    import ctypes as C
    from vampire_amd import _capi
    from vampire_amd.ops import HotPath, voxel_pooling, conv3d_bf16
    lib = _capi.load()
    hp = HotPath(CFG_TINY, dev)
    x = torch.zeros(64, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr())
    # gate + conv: unsupported channel counts, NULL tensors, missing workspace
    assert lib.vamp_gate_conv1x1_supported(34, 10, 80) == 0 and lib.vamp_gate_conv1x1_supported(16, 10, 80) == 1
    assert lib.vamp_gate_conv1x1_forward(1, 34, 10, 16, 80, _capi.VAMP_DENSITY_SDF_LAPLACE, p(x), p(x), p(x), None, p(x), None) < 0
    assert b"not supported" in lib.vamp_last_error()
    assert lib.vamp_gate_conv1x1_forward(1, 4, 2, 16, 8, 7, p(x), p(x), p(x), None, p(x), None) < 0          # density_mode
    assert lib.vamp_gate_conv1x1_backward(1, 4, 2, 16, 8, _capi.VAMP_DENSITY_SIGMOID, p(x), p(x), p(x), p(x), p(x), p(x), p(x),
                                          None, None, 0, None) < 0                                             # no workspace
    # 16-bit conv: dtype must be bf16 or fp16; shape outside the tiles
    d = _capi.VampConvDesc()
    d.B, d.cin, d.cout, d.Z, d.Y, d.X = 1, 16, 16, 2, 4, 8
    assert lib.vamp_conv3d_half_forward(C.byref(d), _capi.VAMP_F32, p(x), p(x), p(x), None) < 0
    d.cin = 40
    assert lib.vamp_conv3d_bf16_supported(C.byref(d)) == 0 and lib.vamp_conv3d_bf16_forward(C.byref(d), p(x), p(x), p(x), None) < 0
    with pytest.raises(TypeError):
        conv3d_bf16(torch.zeros(1, 16, 2, 4, 8, device=dev), torch.zeros(16, 16, 3, 3, 3, device=dev))            # fp32 tensors
    # logits lift: the features must be fp32
    ld = hp.lift_desc(1, 6, 4, _capi.VAMP_BF16)
    assert lib.vamp_lift_forward_logits(C.byref(ld), p(x), p(x), p(x), p(x), p(x), _capi.VAMP_F32, p(x), p(x), p(x), None,
                                        p(x), 64, None) < 0
    # voxel pooling: empty grid, CPU tensors
    pd = _capi.VampPoolDesc(1, 4, 16, 0, 4, 1, _capi.VAMP_F32)
    assert lib.vamp_voxel_pooling_workspace_bytes(C.byref(pd)) == 0
    assert lib.vamp_voxel_pooling_forward(C.byref(pd), p(x), p(x), p(x), p(x), 64, None) < 0
    with pytest.raises(_capi.VampireHipError):
        voxel_pooling(torch.zeros(1, 1, 1, 2, 2, 3, dtype=torch.int32), torch.zeros(1, 1, 1, 2, 2, 4), (4, 4, 1))
    # the library is still usable afterwards
    out = voxel_pooling(torch.zeros(1, 1, 1, 2, 2, 3, dtype=torch.int32, device=dev), torch.ones(1, 1, 1, 2, 2, 4, device=dev), (4, 4, 1))
    assert float(out[0, :, 0, 0].sum()) == 16.0


# --------------------------------------------------------------------------- round 4: the one-kernel camera forward, pinned directly
@pytest.mark.parametrize("name,cfg,golden", [("A", CFG_A, "full_checksums.json"), ("B", CFG_B, "full_checksums.json"),
                                             ("D", None, "cfgd_checksums.json")])
def test_direct_kernel_taps_full_size(dev, name, cfg, golden):
    """The taps of cam_fwd_direct_kernel ITSELF (vamp_render_camera_direct_taps runs the kernel's own
    direct_tap in the kernel's own wave composition), not of its sibling vamp_render_indices: the inside
    mask is the reference's bit for bit (sha256 of all 5.7 M / 23 M samples); the floor taps of the inside
    samples equal the reference's except where the fp64 ray line and the reference's fp32 chain fall on
    two sides of an integer -- there the coordinate is within 1e-4 of that integer (the trilinear sample is
    continuous across it) and such samples are fewer than 1 in 10 000."""
    if cfg is None:
        from vampire_amd.config import CFG_D as cfg
    with open(os.path.join(GOLDEN, golden)) as f:
        ref = json.load(f)[name]
    hp = hot(cfg, dev)
    rm = torch.tensor(ref["render_mats"], dtype=torch.float32, device=dev)
    inside_d, dx, dy, dz, fxyz = hp.render_direct_taps(rm, coords=True)
    assert int(inside_d.sum()) == ref["render_inside_count"] and _sha(inside_d) == ref["render_inside_sha256"]
    inside, rx, ry, rz = hp.render_indices(render_mats=rm)        # hash-equal to the reference (test_full_size_checksums)
    for nm, t in (("ix0", rx), ("iy0", ry), ("iz0", rz)):
        assert _sha(t) == ref[f"render_{nm}_sha256"], nm
    m = inside.bool()
    n_in = int(m.sum())
    for axis, (a, b) in enumerate(((dx, rx), (dy, ry), (dz, rz))):
        bad = m & (a != b)
        nbad = int(bad.sum())
        assert nbad <= 1e-4 * n_in, (axis, nbad, n_in)
        if nbad:
            f = fxyz[..., axis][bad]
            assert float((f - f.round()).abs().max()) <= 1e-4, axis
            assert int((a[bad].int() - b[bad].int()).abs().max()) == 1, axis


def _sample_check(t, ref, key, what, rtol=1e-4, atol=0.0, abs_tol=None):
    """10 000 strided ELEMENTS of a full-size tensor against the reference's (make_golden.strided_sample):
    every element within rtol of the sample's largest magnitude (+ atol) and, with `abs_tol`, within that
    ABSOLUTE error (north_star: rendered depth / semantics within 1e-4 fp32).  Returns (worst error / scale,
    worst absolute error)."""
    want = ref[key].float()
    stride = int(ref[key + "_stride"])
    got = t.detach().float().flatten()[::stride][:want.numel()].cpu()
    assert got.numel() == want.numel(), what
    err = (got - want).abs()
    lim = atol + rtol * float(want.abs().max())
    if abs_tol is not None:
        lim = min(lim, abs_tol)
    assert float(err.max()) <= lim, f"{what}: max element error {float(err.max()):.3e} > {lim:.3e}"
    return float(err.max()) / max(float(want.abs().max()), 1e-30), float(err.max())


@pytest.mark.parametrize("name,cfg,mode", [("A", CFG_A, "sdf"), ("B", CFG_B, "sdf"), ("Bnaive", CFG_B, "naive")])
def test_full_size_elementwise_samples(dev, name, cfg, mode):
    """The DEFAULT path (one-kernel camera forward, fused BEV forward, cell-list backward) at cfg-A / cfg-B,
    element by element: 10 000 strided elements of each of the eight render outputs, the four volume
    gradients, the lift output and its two gradients against the reference run here on CPU
    (tests/golden/full_samples.npz, make_golden.py --samples-only) -- every element within 1e-4 of the
    tensor's scale (north_star's bar), where the block-sum tests only bound averages."""
    ref = load_golden("full_samples.npz")
    with open(os.path.join(GOLDEN, "full_checksums.json")) as f:
        mats = json.load(f)[name[0]]
    with open(os.path.join(GOLDEN, "full_grad_checksums.json")) as f:
        gref = json.load(f)[name[0]]
    cfg = dataclasses.replace(cfg, density_mode=mode)
    hp = hot(cfg, dev)
    # (the default path whatever the environment's switches say: runs of the suite with VAMP_CAM_DIRECT=0 etc.
    # still pin the default here)
    hp.impl.update(cam_direct="auto", bev_fused=True, ert=True, fwd_merged=True)
    lm = torch.tensor(mats["lift_mats"], dtype=torch.float32, device=dev)
    rm = torch.tensor(mats["render_mats"], dtype=torch.float32, device=dev)
    worst = {}
    if mode == "sdf":
        depth, feat = synthetic.lift_inputs(cfg, 1, seed=0, device=dev)
        depth.requires_grad_(True); feat.requires_grad_(True)
        vox = hp.lift(depth, feat, lm)
        g_vox = _upstream([vox.shape], gref["seed_lift"], dev)[0]
        g_vox.view(-1)[gref["lift_upstream_zero_idx"]] = 0.0
        vox.backward(g_vox)
        worst["lift"] = _sample_check(vox, ref, f"{name}_lift", "lift", rtol=1e-5)
        worst["grad_depth"] = _sample_check(depth.grad, ref, f"{name}_grad_depth", "grad_depth")
        worst["grad_feat"] = _sample_check(feat.grad, ref, f"{name}_grad_feat", "grad_feat")
    vols = [v.requires_grad_(True) for v in synthetic.render_inputs(cfg, 1, seed=0, device=dev)]
    beta = torch.tensor(gref["beta"], device=dev, requires_grad=True) if mode == "sdf" else None
    outs = hp.render(*vols, beta, render_mats=rm)
    torch.autograd.backward(outs, _upstream([o.shape for o in outs], 4343 if mode == "sdf" else 4545, dev))
    # the eight rendered tensors: 1e-4 ABSOLUTE per element (depth_preds reaches 70.4, seg logits 2.2: under the
    # scale-relative bound alone they would be allowed 5e-4 / 2e-4); the gradients: 1e-4 of the tensor's scale
    # (the one-kernel camera forward takes its density samples on the reference's own fp32 coordinate chain:
    # depth_preds 4.3e-5 m at cfg-A.  Round 5's second mode -- the samples on the ray's exact line, 2.2e-4 m -- was
    # outside the bar and is gone.)
    for n_, o in zip(NAMES, outs):
        worst[n_] = _sample_check(o, ref, f"{name}_{n_}", n_, abs_tol=1e-4)
    for k, v in zip(("density_feature", "semantic_logits", "base", "rgb"), vols):
        worst["grad_" + k] = _sample_check(v.grad, ref, f"{name}_grad_{k}", "grad_" + k)
    tag = f"cfg-{name}"
    print(f"{tag} worst element error / scale:", {k: f"{v[0]:.1e}" for k, v in worst.items()})
    print(f"{tag} worst ABSOLUTE element error:", {k: f"{v[1]:.1e}" for k, v in worst.items()})


@pytest.mark.parametrize("tag", ["smooth", "nonaffine"])
@pytest.mark.parametrize("direct", [True, False], ids=["direct", "planned"])
def test_render_smooth_and_nonaffine_fixtures(dev, tag, direct):
    """Two more reference fixtures for the camera forward (tests/golden/make_golden.py: make_smooth):
    SMOOTH volumes, where a deviation of the sample coordinates is a bias and does not average out as on
    white noise (the one-kernel forward must stay an order under the 1e-4 bar there), and an `ida` that
    makes get_geometry non-affine in the depth (bv2:334-338) -- the one-kernel forward then marches
    every bin with its own length (render_cam_direct.hip: delta_at)."""
    r = load_golden(f"tiny_render_{tag}.npz")
    cfg = dataclasses.replace(CFG_TINY, density_mode="sdf", cat_seg=False)
    hp = hot(cfg, dev)
    hp.impl["cam_direct"] = direct
    rm = r["render_mats"].to(dev)
    vols = [r[k].to(dev).requires_grad_(True) for k in ("density_feature", "semantic_logits", "base", "rgb")]
    beta = r["beta"].reshape(()).to(dev).requires_grad_(True)
    outs = hp.render(*vols, beta, render_mats=rm)
    tol = 2e-5 if tag == "smooth" else 1e-4
    for n_, o in zip(NAMES, outs):
        close(o, r[n_], atol=tol, rtol=tol, scale="max", what=f"{tag} {n_}")
    torch.autograd.backward(outs, [r["g_" + n].to(dev) for n in NAMES])
    for k, v in zip(("density_feature", "semantic_logits", "base", "rgb"), vols):
        close(v.grad, r["grad_" + k], atol=1e-5, rtol=1e-4, scale="max", what=f"{tag} grad_{k}")
    close(beta.grad.reshape(1), r["grad_beta"], atol=1e-3, rtol=1e-3, what=f"{tag} grad_beta")
    with torch.no_grad():                                  # forward-only call: same values
        outs2 = hp.render(*[v.detach() for v in vols], beta.detach(), render_mats=rm)
    for a, b in zip(outs, outs2):
        close(a, b, atol=1e-6, rtol=1e-6, scale="max", what=f"{tag} no-grad forward")
