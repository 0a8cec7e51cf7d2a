"""Deterministic synthetic Gaussian scenes for tests and the benchmark.

Inputs are in the layout ``render()`` feeds the rasterizer with
(gaussian_renderer/__init__.py:54-83): means [P,3], SH [P,M,3], opacity [P,1],
scales [P,3] (already activated), unit quaternions [P,4] (w,x,y,z).  Means are
uniform in the cube the reference uses for random initialisation
(scene/dataset_readers.py:280).
"""
import math

import torch

from .cameras import make_camera, orbit_eye

FOVX = 0.6911112  # transforms_video/rotate2/transforms_video.json:2


class Scene:
    def __init__(self, means3D, shs, opacities, scales, rotations, sh_degree):
        self.means3D = means3D
        self.shs = shs
        self.opacities = opacities
        self.scales = scales
        self.rotations = rotations
        self.sh_degree = sh_degree

    def to(self, device):
        return Scene(self.means3D.to(device), self.shs.to(device), self.opacities.to(device), self.scales.to(device),
                     self.rotations.to(device), self.sh_degree)


def make_scene(P, sh_degree=0, log_scale=(-5.3, -4.3), extent=(1.3, 1.3, 1.3), seed=1234, sh_coeffs=None):
    """``sh_coeffs`` is M, the number of SH coefficients held per Gaussian
    (defaults to (sh_degree+1)^2; the reference's model always holds 16)."""
    g = torch.Generator().manual_seed(seed)
    M = (sh_degree + 1) ** 2 if sh_coeffs is None else sh_coeffs
    ext = torch.tensor(extent, dtype=torch.float32)
    means = (torch.rand(P, 3, generator=g) * 2.0 - 1.0) * ext
    q = torch.randn(P, 4, generator=g)
    q = q / q.norm(dim=1, keepdim=True)
    opac = 0.1 + 0.85 * torch.rand(P, 1, generator=g)
    shs = torch.zeros(P, M, 3)
    shs[:, 0, :] = (torch.rand(P, 3, generator=g) - 0.5) / 0.28209479177387814
    if M > 1:
        shs[:, 1:, :] = 0.1 * torch.randn(P, M - 1, 3, generator=g)
    lo, hi = log_scale
    scales = torch.exp(lo + (hi - lo) * torch.rand(P, 3, generator=g))
    return Scene(means.contiguous(), shs.contiguous(), opac.contiguous(), scales.contiguous(), q.contiguous(),
                 sh_degree)


def make_view(width, height, index=0, count=8, radius=4.0, device="cpu"):
    return make_camera(width, height, FOVX, orbit_eye(index, count, radius), device=device)


# name -> (P, W, H, sh_degree, M, log_scale, extent): the benchmark configurations
CONFIGS = {
    "c1": (256, 64, 64, 0, 1, (-3.0, -2.0), (1.3, 1.3, 1.3)),
    "c2": (100_000, 800, 800, 0, 1, (-5.3, -4.3), (1.3, 1.3, 1.3)),
    "c3": (300_000, 800, 800, 3, 16, (-5.6, -4.6), (1.3, 1.3, 1.3)),
    "c4": (3_000_000, 1920, 1080, 3, 16, (-6.2, -5.2), (1.3, 0.7, 1.3)),
    "c5": (6_000_000, 1920, 1080, 3, 16, (-6.5, -5.5), (1.3, 0.7, 1.3)),
}


def make_config(name, seed=1234):
    P, W, H, D, M, ls, ext = CONFIGS[name]
    return make_scene(P, D, ls, ext, seed, M), W, H


def tan_half_fov(cam):
    return math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5)
