"""Minimal Gaussian parameter store with the read interface ``render()`` uses.

Holds raw (pre-activation) parameters as ``nn.Parameter`` and applies the
reference model's activations (scene/gaussian_model.py:29-45,113-144): ``exp``
on scales, ``sigmoid`` on opacity, L2 normalisation on quaternions, DC and
higher SH bands concatenated.  Densification, optimiser set-up and PLY I/O of
the reference model are outside the rasterizer path and are not provided here.
"""
import torch
from torch import nn


def quaternion_to_rotation(q):
    """[N,4] (w,x,y,z), normalised here, -> [N,3,3] (utils/general_utils.py:78-99)."""
    q = q / q.norm(dim=1, keepdim=True)
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    rows = [1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
            2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
            2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)]
    return torch.stack(rows, dim=1).view(-1, 3, 3)


def covariance_from_scaling_rotation(scaling, scaling_modifier, rotation):
    """Upper triangle (xx,xy,xz,yy,yz,zz) of R S S^T R^T (scene/gaussian_model.py:29-33)."""
    L = quaternion_to_rotation(rotation) * (scaling_modifier * scaling).unsqueeze(1)
    cov = L @ L.transpose(1, 2)
    return torch.stack([cov[:, 0, 0], cov[:, 0, 1], cov[:, 0, 2], cov[:, 1, 1], cov[:, 1, 2], cov[:, 2, 2]], dim=1)


def inverse_sigmoid(x):
    return torch.log(x / (1 - x))


class GaussianModel:
    def __init__(self, sh_degree):
        self.active_sh_degree = 0
        self.max_sh_degree = sh_degree
        self._xyz = self._features_dc = self._features_rest = torch.empty(0)
        self._scaling = self._rotation = self._opacity = torch.empty(0)

    @classmethod
    def from_activated(cls, means3D, shs, opacities, scales, rotations, sh_degree, active_sh_degree=None,
                       device=None):
        """Build a model whose activated properties reproduce the given tensors."""
        m = cls(sh_degree)
        m.active_sh_degree = sh_degree if active_sh_degree is None else active_sh_degree
        dev = device if device is not None else means3D.device
        mk = lambda t: nn.Parameter(t.detach().clone().to(dev).contiguous().requires_grad_(True))
        m._xyz = mk(means3D)
        m._features_dc = mk(shs[:, :1, :])
        m._features_rest = mk(shs[:, 1:, :])
        m._scaling = mk(torch.log(scales))
        m._rotation = mk(rotations)
        m._opacity = mk(inverse_sigmoid(opacities))
        return m

    def parameters(self):
        return [self._xyz, self._features_dc, self._features_rest, self._opacity, self._scaling, self._rotation]

    @property
    def get_xyz(self):
        return self._xyz

    @property
    def get_scaling(self):
        return torch.exp(self._scaling)

    @property
    def get_rotation(self):
        return torch.nn.functional.normalize(self._rotation)

    @property
    def get_opacity(self):
        return torch.sigmoid(self._opacity)

    @property
    def get_features(self):
        return torch.cat((self._features_dc, self._features_rest), dim=1)

    def get_covariance(self, scaling_modifier=1):
        return covariance_from_scaling_rotation(self.get_scaling, scaling_modifier, self._rotation)

    def oneupSHdegree(self):
        if self.active_sh_degree < self.max_sh_degree:
            self.active_sh_degree += 1
