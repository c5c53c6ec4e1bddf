"""Shape/bound configuration of the lift+render hot path.

`PathConfig` carries exactly the subset of the reference's ``backbone_conf``
(/root/reference/src/exps/nuscenes/base_exp.py:40-92) that determines shapes
and constants on the hot path.  The named presets are the configurations of
BASELINE.json / SURVEY.md §8(d).
"""
from dataclasses import dataclass, field, asdict
from typing import Tuple, List


def axis_cells(bound) -> int:
    """Number of cells along one axis.

    Mirrors the reference's ``int((hi - lo) / step)`` truncation
    (base_vampire2.py:276-281) -- including its float quirk, e.g.
    ``int((2.0 - -0.4) / 0.8) == 2``.
    """
    lo, hi, step = bound
    return int((hi - lo) / step)


@dataclass(frozen=True)
class PathConfig:
    x_bound_seg: Tuple[float, float, float] = (-51.2, 51.2, 0.4)
    y_bound_seg: Tuple[float, float, float] = (-51.2, 51.2, 0.4)
    z_bound_seg: Tuple[float, float, float] = (-5.0, 3.0, 0.4)
    x_bound_det: Tuple[float, float, float] = (-51.2, 51.2, 0.4)
    y_bound_det: Tuple[float, float, float] = (-51.2, 51.2, 0.4)
    z_bound_det: Tuple[float, float, float] = (-1.0, 3.0, 0.4)
    d_bound: Tuple[float, float, float] = (2.0, 70.4, 0.8)
    final_dim: Tuple[int, int] = (256, 704)
    downsample_factor: int = 4
    mid_channels: int = 16
    num_classes: int = 18
    num_cams: int = 6
    density_mode: str = "sdf"
    sdf_bias: float = -1.0
    cat_seg: bool = False

    # ---- derived shapes -------------------------------------------------
    @property
    def fH(self) -> int:
        return self.final_dim[0] // self.downsample_factor

    @property
    def fW(self) -> int:
        return self.final_dim[1] // self.downsample_factor

    @property
    def D(self) -> int:
        """Depth planes = len(arange(*d_bound)) (base_vampire2.py:258)."""
        import math
        lo, hi, step = self.d_bound
        return int(math.ceil((hi - lo) / step))

    @property
    def vX(self) -> int:
        return axis_cells(self.x_bound_seg)

    @property
    def vY(self) -> int:
        return axis_cells(self.y_bound_seg)

    @property
    def vZ(self) -> int:
        return axis_cells(self.z_bound_seg)

    @property
    def oX(self) -> int:
        return axis_cells(self.x_bound_det)

    @property
    def oY(self) -> int:
        return axis_cells(self.y_bound_det)

    @property
    def oZ(self) -> int:
        return axis_cells(self.z_bound_det)

    def to_dict(self):
        return asdict(self)

    # ---- algorithmic bytes (SURVEY.md §8d / BASELINE.md §5), per sample --
    def algorithmic_bytes(self, in_bytes: int = 4) -> dict:
        P = self.num_cams * self.fH * self.fW
        V = self.vZ * self.vY * self.vX
        C, K, D = self.mid_channels, self.num_classes, self.D
        YX = self.oY * self.oX
        lift_fwd = in_bytes * P * (D + C) + 4 * C * V
        render_out = 4 * P * (K + 4) + 4 * YX * (K + 4) + 4 * self.oZ * YX * (1 + C)
        render_fwd = in_bytes * (1 + K + 3 + C) * V + render_out
        lift_bwd = 4 * C * V + 2 * 4 * P * (D + C)
        render_bwd = render_out + 2 * 4 * (1 + K + 3 + C) * V
        return dict(lift_fwd=lift_fwd, render_fwd=render_fwd,
                    lift_bwd=lift_bwd, render_bwd=render_bwd,
                    fwd=lift_fwd + render_fwd,
                    fwd_bwd=lift_fwd + render_fwd + lift_bwd + render_bwd)


# cfg-A: the reference's default experiment (base_exp.py:40-63)
CFG_A = PathConfig()

# cfg-B: BASELINE.json configs[1], "200x200x16 voxel grid" (= the Occ3D range
# hard-coded at base_vampire2.py:295)
CFG_B = PathConfig(x_bound_seg=(-40.0, 40.0, 0.4), y_bound_seg=(-40.0, 40.0, 0.4),
                   z_bound_seg=(-1.0, 5.4, 0.4),
                   x_bound_det=(-40.0, 40.0, 0.4), y_bound_det=(-40.0, 40.0, 0.4),
                   z_bound_det=(-1.0, 3.0, 0.4))

# cfg-D: BASELINE.json configs[3], 512x1408 input, 400x400x32 grid (0.2 m)
CFG_D = PathConfig(x_bound_seg=(-40.0, 40.0, 0.2), y_bound_seg=(-40.0, 40.0, 0.2),
                   z_bound_seg=(-1.0, 5.4, 0.2),
                   x_bound_det=(-40.0, 40.0, 0.2), y_bound_det=(-40.0, 40.0, 0.2),
                   z_bound_det=(-1.0, 3.0, 0.2),
                   final_dim=(512, 1408))

# tiny: the golden-fixture configuration
CFG_TINY = PathConfig(x_bound_seg=(-6.4, 6.4, 0.8), y_bound_seg=(-6.4, 6.4, 0.8),
                      z_bound_seg=(-2.0, 2.0, 0.8),
                      x_bound_det=(-6.4, 6.4, 0.8), y_bound_det=(-6.4, 6.4, 0.8),
                      z_bound_det=(-0.4, 2.0, 0.8),
                      d_bound=(2.0, 18.4, 0.8), final_dim=(32, 88),
                      mid_channels=4, num_classes=5)

PRESETS = {"A": CFG_A, "B": CFG_B, "D": CFG_D, "tiny": CFG_TINY}
