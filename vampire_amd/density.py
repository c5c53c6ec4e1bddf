"""SDF -> density activation with a learnable scale (host-side module; the renderer applies
the same function in-kernel).  Follows /root/reference/src/utils/render_utils.py:30-46:
``sigma(s) = (1/b) * (0.5 + 0.5 * sign(s - bias) * expm1(-|s - bias| / b))``, ``b = |beta| + beta_min``.
The parameter is named ``beta`` so that reference checkpoints (``density.beta``) load."""
import torch
from torch import nn


class ModifyLaplaceDensity(nn.Module):
    def __init__(self, beta=0.1, bias=5.0, beta_min=0.0001):
        super().__init__()
        self.beta = nn.Parameter(torch.tensor(beta))
        self.beta_min = beta_min
        self.bias = bias

    def get_beta(self):
        return self.beta.abs() + self.beta_min

    def forward(self, sdf, beta=None):
        b = self.get_beta() if beta is None else beta
        t = sdf - self.bias
        return (1.0 / b) * (0.5 + 0.5 * t.sign() * torch.expm1(-t.abs() / b))
