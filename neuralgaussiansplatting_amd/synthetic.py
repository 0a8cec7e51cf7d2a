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


def make_clustered_scene(P, sh_degree=0, log_scale_mean=-5.6, log_scale_sigma=1.0, extent=(1.3, 0.7, 1.3), seed=1234,
                         sh_coeffs=None, clusters=200, clustered=0.7, max_log_scale=-1.2):
    """A stand-in for a trained scene (the reference trains on COLMAP captures, scene/dataset_readers.py:149-301,
    none of which is available offline): ``clustered`` of the Gaussians sit in ``clusters`` anisotropic blobs of
    very unequal population, the rest is uniform in the box; sizes are log-normal (sigma 1: a tail of splats
    hundreds of pixels wide next to a bulk of two or three pixels) with a per-axis spread on top, so shapes range
    from discs to needles; opacities are bimodal (nearly transparent / nearly opaque).  Tiles end up very unevenly
    loaded, which the uniform scenes of ``make_scene`` never are."""
    g = torch.Generator().manual_seed(seed)
    M = (sh_degree + 1) ** 2 if sh_coeffs is None else sh_coeffs
    ext = torch.tensor(extent, dtype=torch.float32)
    n_c = int(P * clustered)
    centres = (torch.rand(clusters, 3, generator=g) * 2.0 - 1.0) * ext * 0.9
    axes = torch.exp(-3.6 + 2.2 * torch.rand(clusters, 3, generator=g))  # 0.027 .. 0.25 per axis
    rot, _ = torch.linalg.qr(torch.randn(clusters, 3, 3, generator=g))
    weight = torch.exp(1.2 * torch.randn(clusters, generator=g))
    member = torch.multinomial(weight / weight.sum(), n_c, replacement=True, generator=g)
    local = torch.randn(n_c, 3, generator=g) * axes[member]
    blob = centres[member] + torch.bmm(local.unsqueeze(1), rot[member]).squeeze(1)
    uniform = (torch.rand(P - n_c, 3, generator=g) * 2.0 - 1.0) * ext
    means = torch.cat([blob, uniform])[torch.randperm(P, generator=g)]
    q = torch.randn(P, 4, generator=g)
    q = q / q.norm(dim=1, keepdim=True)
    low = torch.rand(P, 1, generator=g) < 0.45
    opac = torch.sigmoid(torch.where(low, -2.5 + 0.8 * torch.randn(P, 1, generator=g),
                                     2.5 + 1.0 * torch.randn(P, 1, generator=g)))
    shs = torch.zeros(P, M, 3)
    shs[:, 0, :] = (torch.rand(P, 3, generator=g) - 0.5) / 0.28209479177387814
    if M > 1:
        shs[:, 1:, :] = 0.1 * torch.randn(P, M - 1, 3, generator=g)
    size = log_scale_mean + log_scale_sigma * torch.randn(P, 1, generator=g)
    scales = torch.exp(torch.clamp(size + 0.6 * torch.randn(P, 3, generator=g), max=max_log_scale))
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


# name -> (P, W, H, sh_degree, M, mean of the log-size, extent): the headline configuration's counts and image with
# the shape of a trained scene (make_clustered_scene), and a reduced copy for tests that run it several times
CLUSTERED_CONFIGS = {
    "c4s": (3_000_000, 1920, 1080, 3, 16, -6.6, (1.3, 0.7, 1.3)),
    "c4s_small": (200_000, 640, 360, 3, 16, -5.6, (1.3, 0.7, 1.3)),
}


def make_config(name, seed=1234):
    if name in CLUSTERED_CONFIGS:
        P, W, H, D, M, mean, ext = CLUSTERED_CONFIGS[name]
        return make_clustered_scene(P, D, mean, 1.0, ext, seed, M), W, H
    P, W, H, D, M, ls, ext = CONFIGS[name]
    return make_scene(P, D, ls, ext, seed, M), W, H


def tan_half_fov(cam):
    return math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5)
