"""Real spherical-harmonics evaluation in PyTorch, degrees 0-3.

Used by ``render()`` when the SH->RGB conversion is done outside the
rasterizer (the reference's ``convert_SHs_python`` branch,
gaussian_renderer/__init__.py:74-79, which calls utils/sh_utils.py:57-112).
Same basis ordering and constants as the rasterizer kernel.
"""
import torch

C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
      1.445305721320277, -0.5900435899266435)


def sh_basis(deg, dirs):
    """[..., (deg+1)^2] basis values at unit directions ``dirs`` [..., 3]."""
    x, y, z = dirs[..., 0], dirs[..., 1], dirs[..., 2]
    out = [torch.full_like(x, C0)]
    if deg > 0:
        out += [-C1 * y, C1 * z, -C1 * x]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        out += [C2[0] * xy, C2[1] * yz, C2[2] * (2.0 * zz - xx - yy), C2[3] * xz, C2[4] * (xx - yy)]
    if deg > 2:
        out += [C3[0] * y * (3 * xx - yy), C3[1] * xy * z, C3[2] * y * (4 * zz - xx - yy),
                C3[3] * z * (2 * zz - 3 * xx - 3 * yy), C3[4] * x * (4 * zz - xx - yy), C3[5] * z * (xx - yy),
                C3[6] * x * (xx - 3 * yy)]
    return torch.stack(out, dim=-1)


def eval_sh(deg, sh, dirs):
    """``sh`` [..., C, K] with K >= (deg+1)^2, ``dirs`` [..., 3] unit vectors -> [..., C]."""
    if not 0 <= deg <= 3:
        raise ValueError("SH degree must be in 0..3")
    n = (deg + 1) ** 2
    if sh.shape[-1] < n:
        raise ValueError("not enough SH coefficients for degree %d" % deg)
    basis = sh_basis(deg, dirs)
    return (sh[..., :n] * basis.unsqueeze(-2)).sum(dim=-1)


def RGB2SH(rgb):
    return (rgb - 0.5) / C0


def SH2RGB(sh):
    return sh * C0 + 0.5
