"""The drop-in module keeps the reference's BaseVAMPIRE2 interface: constructor
kwargs, buffer names/shapes/values, submodule names, method names.  CPU-only checks; the forward
itself needs the GPU (tests/test_hip_parity.py::test_backbone_forward_backward)."""
import inspect

import pytest
import torch

from vampire_amd.backbone import BaseVAMPIRE2
from vampire_amd.config import CFG_TINY

KW = dict(x_bound_seg=[-51.2, 51.2, 0.4], y_bound_seg=[-51.2, 51.2, 0.4], z_bound_seg=[-5., 3., 0.4],
          x_bound_det=[-51.2, 51.2, 0.4], y_bound_det=[-51.2, 51.2, 0.4], z_bound_det=[-1., 3., 0.4],
          d_bound=[2.0, 70.4, 0.8], final_dim=(256, 704), downsample_factor=4, upsample_factor=4,
          mid_channels=16, output_channels=80, img_backbone_conf=dict(type="ResNet", depth=50),
          img_neck_conf=dict(type="SECONDFPN", out_channels=[128, 128, 128, 128]), num_classes=18,
          density_mode="sdf", sdf_bias=-1.0, cat_pos=True, cat_seg=False)


@pytest.fixture(scope="module")
def module():
    torch.manual_seed(0)
    return BaseVAMPIRE2(**KW)


def test_constructor_signature_matches_reference():
    names = list(inspect.signature(BaseVAMPIRE2.__init__).parameters)[1:]
    assert names == ["x_bound_seg", "y_bound_seg", "z_bound_seg", "x_bound_det", "y_bound_det",
                     "z_bound_det", "d_bound", "final_dim", "downsample_factor", "upsample_factor",
                     "mid_channels", "output_channels", "img_backbone_conf", "img_neck_conf",
                     "num_classes", "density_mode", "sdf_bias", "cat_pos", "cat_seg", "use_da"]


def test_buffers_and_attributes(module):
    bufs = dict(module.named_buffers(recurse=False))
    shapes = {"frustum": (86, 64, 176, 4), "camera_mids": (85,), "bev_mids": (10,),
              "voxel_coords": (20, 256, 256, 4), "occ_coords": (200, 200, 16, 3),
              "norm_voxel_coords": (20, 256, 256, 3), "output_coords": (10, 256, 256, 4)}
    for name, shp in shapes.items():
        assert tuple(bufs[name].shape) == shp, name
    assert (module.fD, module.fH, module.fW) == (85, 64, 176)
    assert (module.vZ, module.vY, module.vX, module.oY, module.depth_channels) == (20, 256, 256, 256, 86)
    # values: first/last frustum column, camera mids, flipped bev mids, voxel centres
    assert float(module.frustum[0, 0, -1, 0]) == 703.0 and float(module.frustum[-1, 0, 0, 2]) == pytest.approx(70.0)
    assert float(module.camera_mids[0]) == pytest.approx(2.4)
    assert float(module.bev_mids[0]) == pytest.approx(2.8) and float(module.bev_mids[-1]) == pytest.approx(-0.8)
    assert module.voxel_coords[0, 0, 0].tolist() == pytest.approx([-51.0, -51.0, -4.8, 1.0])
    assert module.occ_coords[0, 0, 0].tolist() == pytest.approx([-39.8, -39.8, -0.8])


def test_submodules_and_state_dict_names(module):
    for name in ("img_backbone", "img_neck", "mapping_along_depth", "channel_lower", "base_conv",
                 "density_conv", "seg_conv", "density", "rgb_conv", "voxel_output", "upsample2d"):
        assert hasattr(module, name), name
    keys = set(module.state_dict().keys())
    for k in ("density.beta", "mapping_along_depth.0.weight", "channel_lower.weight",
              "base_conv.init_dres.weight", "base_conv.hg1.conv1.0.weight", "base_conv.hg2.conv6.0.weight",
              "density_conv.bias", "seg_conv.weight", "rgb_conv.0.weight", "voxel_output.0.weight"):
        assert k in keys, k
    # in-repo layers around the path: 777 111 parameters in the reference (collectives)
    n = sum(p.numel() for nme, p in module.named_parameters()
            if not nme.startswith(("img_backbone", "img_neck")))
    assert n == 777111
    assert float(module.density_conv.bias[0]) == pytest.approx(-11.0)


def test_methods_exist_with_reference_signatures(module):
    sig = lambda f: list(inspect.signature(f).parameters)
    assert sig(module.get_geometry) == ["sensor2ego_mat", "intrin_mat", "ida_mat", "bda_mat"]
    assert sig(module.get_pixel) == ["sensor2ego_mat", "intrin_mat", "ida_mat", "bda_mat"]
    assert sig(module.get_voxel_feats) == ["frustum_feats", "sweep_index", "mats_dict", "clamp_extreme"]
    assert sig(module.volume_rendering_from_multiple_views) == \
        ["geom_xyz", "density_feature", "semantic_logits", "voxel_features", "rgb"]
    assert sig(module.forward) == ["sweep_imgs", "mats_dict", "inrange_pts", "timestamps"]
    with pytest.raises(NotImplementedError):
        module.forward(torch.zeros(1, 2, 6, 3, 8, 8), {})


def test_get_pixel_matches_oracle():
    from oracle import aten_oracle as O
    from vampire_amd import synthetic
    from vampire_amd.geometry import PathGeometry
    c = CFG_TINY
    m = BaseVAMPIRE2(**{**KW, **dict(x_bound_seg=list(c.x_bound_seg), y_bound_seg=list(c.y_bound_seg),
                                     z_bound_seg=list(c.z_bound_seg), x_bound_det=list(c.x_bound_det),
                                     y_bound_det=list(c.y_bound_det), z_bound_det=list(c.z_bound_det),
                                     d_bound=list(c.d_bound), final_dim=c.final_dim, mid_channels=4,
                                     num_classes=5, cat_pos=False,
                                     img_neck_conf=dict(out_channels=[8] * 4))})
    s2e, K, ida = synthetic.camera_rig(c, 2, jitter=2.0, seed=1)
    bda = synthetic.bda_matrix(2, rot_deg=12.0)
    got = m.get_pixel(s2e, K, ida, bda)
    want = O.ego_to_pixel(PathGeometry(c).voxel_coords, s2e, K, ida, bda)
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-4)


# --------------------------------------------------------------------------- the three sibling backbones
# numbers below were read off the reference classes themselves (src/layers/backbones/base_lss_impaintor.py:79,
# base_lss.py:16, base_bilinear.py:80) constructed with KW in the build container
SIBLINGS = {
    "BaseLSSImpaintor": dict(params=791511, cat_pos=True, cat_seg=True, occ="norm_occ_coords", unet=True, depth=True),
    "BaseLSS": dict(params=515047, cat_pos=True, cat_seg=True, occ="norm_occ_coords", unet=False, depth=True),
    "BaseBiLinear": dict(params=109991, cat_pos=False, cat_seg=False, occ="norm_occ_coords", unet=False, depth=False),
}


@pytest.mark.parametrize("name", sorted(SIBLINGS))
def test_sibling_backbones_keep_the_reference_interface(name):
    import vampire_amd.backbone as BB
    want = SIBLINGS[name]
    cls = getattr(BB, name)
    kw = {k: v for k, v in KW.items() if k not in ("cat_pos", "cat_seg")}       # class defaults apply
    assert list(inspect.signature(cls.__init__).parameters)[1:] == \
        list(inspect.signature(BaseVAMPIRE2.__init__).parameters)[1:]
    m = cls(**kw)
    assert (m.cat_pos, m.cat_seg) == (want["cat_pos"], want["cat_seg"])
    bufs = dict(m.named_buffers(recurse=False))
    assert want["occ"] in bufs and "occ_coords" not in bufs
    assert tuple(bufs["norm_occ_coords"].shape) == (200, 200, 16, 3)
    # (c - lo) / span * 2 - 1 of the first Occ3D voxel centre (-39.8, -39.8, -0.8) by the seg bounds
    assert bufs["norm_occ_coords"][0, 0, 0].tolist() == pytest.approx(
        [(-39.8 + 51.2) / 102.4 * 2 - 1, (-39.8 + 51.2) / 102.4 * 2 - 1, (-0.8 + 5.0) / 8.0 * 2 - 1], abs=1e-6)
    assert ("norm_voxel_coords" in bufs) == want["cat_pos"]
    n = sum(p.numel() for k, p in m.named_parameters() if not k.startswith(("img_backbone", "img_neck")))
    assert n == want["params"]
    keys = set(m.state_dict())
    assert ("base_conv.hg1.conv1.0.weight" in keys) == want["unet"]
    assert ("base_conv.0.bias" in keys) == (not want["unet"])
    assert ("mapping_along_depth.0.weight" in keys) == want["depth"]
    assert ("feature_conv.weight" in keys) == (not want["depth"])
    sig = lambda f: list(inspect.signature(f).parameters)
    assert sig(m.get_voxel_feats) == ["frustum_feats", "sweep_index", "mats_dict", "clamp_extreme"]
    assert sig(m.forward) == ["sweep_imgs", "mats_dict", "inrange_pts", "timestamps"]
