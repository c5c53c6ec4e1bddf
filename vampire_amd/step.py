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

    def __init__(self, cfg: PathConfig, batch: int, device, seed: int = 0, dtype=torch.float32):
        s2e, K, ida = synthetic.camera_rig(cfg, batch, jitter=1.0 if batch > 1 else 0.0, seed=seed)
        bda = synthetic.bda_matrix(batch)
        self.mats_host = (s2e, K, ida, bda)
        self.lift_mats = lift_matrices(s2e, K, ida, bda).to(device)
        self.render_mats = render_matrices(s2e, K, ida, bda).to(device)
        self.depth, self.feat = synthetic.lift_inputs(cfg, batch, seed=seed, device=device, dtype=dtype)
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
