"""One data-parallel training step of the hot path: lift fwd -> render fwd -> backward.

`LiftRenderStep` is the unit bench.py times and the DDP harness replicates.  It
owns the path's single learnable parameter (the Laplace-density ``beta``,
/root/reference/src/utils/render_utils.py:30-46) so that a DistributedDataParallel
wrapper has a real gradient to all-reduce over RCCL; everything else on the path
is activation-to-activation.
"""
import torch
from torch import nn

from .config import PathConfig
from .geometry import lift_matrices, render_matrices
from .ops import HotPath
from . import synthetic


class LiftRenderStep(nn.Module):
    def __init__(self, cfg: PathConfig, device, hot_path=None):
        super().__init__()
        self.cfg = cfg
        # hot_path is injectable so that the multi-process harness can be exercised on CPU (gloo)
        # in tests with a stand-in; the product always uses the HIP HotPath.
        self.hp = HotPath(cfg, device) if hot_path is None else hot_path
        self.beta = nn.Parameter(torch.tensor(0.1, device=device))      # ModifyLaplaceDensity(beta=0.1)

    def forward(self, depth, feat, vols, lift_mats, render_mats):
        vox = self.hp.lift(depth, feat, lift_mats)
        outs = self.hp.render(*vols, self.beta if self.cfg.density_mode == "sdf" else None,
                              render_mats=render_mats)
        return vox, outs


class SyntheticBatch:
    """Seeded device-resident inputs + fixed upstream gradients for one rank."""

    def __init__(self, cfg: PathConfig, batch: int, device, seed: int = 0, dtype=torch.float32, feat_channel_last=True):
        s2e, K, ida = synthetic.camera_rig(cfg, batch, jitter=1.0 if batch > 1 else 0.0, seed=seed)
        bda = synthetic.bda_matrix(batch)
        self.mats_host = (s2e, K, ida, bda)
        self.lift_mats = lift_matrices(s2e, K, ida, bda).to(device)
        self.render_mats = render_matrices(s2e, K, ida, bda).to(device)
        self.depth, self.feat = synthetic.lift_inputs(cfg, batch, seed=seed, device=device, dtype=dtype)
        if feat_channel_last and dtype == torch.float32:
            # the same [B, N, C, fH, fW] values in the memory layout a torch.channels_last `channel_lower` convolution
            # (bv2:551-553) emits, [B, N, fH, fW, C]: what the lift takes zero-copy (no transposing first launch)
            self.feat = self.feat.permute(0, 1, 3, 4, 2).contiguous().permute(0, 1, 4, 2, 3)
        self.vols = list(synthetic.render_inputs(cfg, batch, seed=seed, device=device, dtype=dtype))
        self.depth.requires_grad_(True)
        self.feat.requires_grad_(True)
        for v in self.vols:
            v.requires_grad_(True)
        self._grads = None

    def upstream(self, vox, outs):
        """Fixed pseudo-loss gradients (generated once, reused every step)."""
        if self._grads is None:
            g = torch.Generator(device=vox.device).manual_seed(1234)
            self._grads = [torch.randn(t.shape, device=t.device, generator=g) * 1e-3
                           for t in (vox,) + tuple(outs)]
        return self._grads

    def zero_grads(self):
        self.depth.grad = None
        self.feat.grad = None
        for v in self.vols:
            v.grad = None


def train_step(model, batch: SyntheticBatch):
    """fwd + bwd of lift+render; gradients land in batch tensors and model.beta."""
    batch.zero_grads()
    vox, outs = model(batch.depth, batch.feat, batch.vols, batch.lift_mats, batch.render_mats)
    torch.autograd.backward((vox,) + tuple(outs), batch.upstream(vox, outs))
    if hasattr(model, "finish"):          # dist.GradSync: join the gradient all-reduce
        model.finish()
    return vox, outs


# ---------------------------------------------------------------------------------------------
# the same step wrapped with the in-repo layers of the reference backbone
# ---------------------------------------------------------------------------------------------
class LayeredStep(nn.Module):
    """The lift + render operators between the reference backbone's own layers -- mapping_along_depth,
    channel_lower, the 3-D UNet (base_conv), the density / semantic / rgb heads, voxel_output
    (bv2:167-210; 777 111 parameters at the reference's configuration: a 3.1 MB gradient bucket) -- so
    that a data-parallel step has a real all-reduce to overlap with its backward (base_cli.py:72, 84,
    105: DDP over the whole model).  The image encoder is not part of it: the step starts from
    synthetic neck features.  `hot_path` is injectable for the CPU (gloo) tests."""

    NECK_CHANNELS = (128, 128, 128, 128)          # base_exp.py:71-80: SECONDFPN out_channels

    def __init__(self, cfg: PathConfig, device, hot_path=None, output_channels=80, occupancy=True):
        super().__init__()
        from .backbone import BaseVAMPIRE2
        bb = BaseVAMPIRE2(
            x_bound_seg=list(cfg.x_bound_seg), y_bound_seg=list(cfg.y_bound_seg), z_bound_seg=list(cfg.z_bound_seg),
            x_bound_det=list(cfg.x_bound_det), y_bound_det=list(cfg.y_bound_det), z_bound_det=list(cfg.z_bound_det),
            d_bound=list(cfg.d_bound), final_dim=tuple(cfg.final_dim), downsample_factor=cfg.downsample_factor,
            upsample_factor=cfg.downsample_factor, mid_channels=cfg.mid_channels, output_channels=output_channels,
            img_backbone_conf=dict(), img_neck_conf=dict(out_channels=list(self.NECK_CHANNELS)),
            num_classes=cfg.num_classes, density_mode=cfg.density_mode, sdf_bias=cfg.sdf_bias, cat_pos=True,
            cat_seg=cfg.cat_seg)
        # the image encoder is out of scope (and would dominate the bucket): drop its parameters
        bb.img_backbone = nn.Identity()
        bb.img_neck = nn.Identity()
        with torch.no_grad():
            bb.density_conv.bias.fill_(cfg.sdf_bias)      # densities in an informative range (the init value saturates every ray)
        self.backbone = bb.to(device)
        if hot_path is not None:
            self.backbone._hot = hot_path
        self.occupancy = occupancy

    @property
    def hp(self):
        return self.backbone.hot_path()

    def forward(self, neck_feats, mats_dict):
        return self.backbone._sweep_from_feats(0, neck_feats, mats_dict, occupancy=self.occupancy)


class LayeredBatch:
    """Seeded neck features [B, 1, N, 512, fH, fW] + the matrices of the synthetic rig, fixed upstream
    gradients for the tensor outputs."""

    def __init__(self, cfg: PathConfig, batch: int, device, seed: int = 0):
        s2e, K, ida = synthetic.camera_rig(cfg, batch, jitter=1.0 if batch > 1 else 0.0, seed=seed)
        bda = synthetic.bda_matrix(batch)
        g = torch.Generator().manual_seed(seed + 77)
        c = sum(LayeredStep.NECK_CHANNELS)
        self.feats = (0.5 * torch.randn(batch, 1, cfg.num_cams, c, cfg.fH, cfg.fW, generator=g)).to(device)
        self.mats = {k: v.to(device) for k, v in dict(
            sensor2ego_mats=s2e[:, None], intrin_mats=K[:, None], ida_mats=ida[:, None],
            sensor2sensor_mats=torch.eye(4).expand(batch, 1, cfg.num_cams, 4, 4).contiguous(), bda_mat=bda).items()}
        self._grads = None

    def upstream(self, outs):
        if self._grads is None:
            g = torch.Generator(device=outs[0].device).manual_seed(4321)
            self._grads = [torch.randn(t.shape, device=t.device, generator=g) * 1e-3 for t in outs]
        return self._grads


def layered_step(model, batch: LayeredBatch):
    """forward + backward of the layered step (model: LayeredStep, possibly under DDP / GradSync);
    parameter gradients land in the module."""
    out = model(batch.feats, batch.mats)
    outs = [t for t in out if torch.is_tensor(t)]
    torch.autograd.backward(outs, batch.upstream(outs))
    if hasattr(model, "finish"):
        model.finish()
    return outs
