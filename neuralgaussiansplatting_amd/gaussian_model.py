"""Minimal Gaussian parameter store with the read interface ``render()`` uses.

Holds raw (pre-activation) parameters as ``nn.Parameter`` and applies the
reference model's activations (scene/gaussian_model.py:29-45,113-144): ``exp``
on scales, ``sigmoid`` on opacity, L2 normalisation on quaternions, DC and
higher SH bands concatenated.  Also the two ends of a run from a point cloud:
``create_from_pcd`` (initial scales from the nearest-neighbour distances,
:150-183) and ``save_ply`` / ``load_ply`` in the reference's file layout
(:289-392).  Densification and the optimiser live with the training step (``train_step.GaussianTrainer``).
"""
import os

import numpy as np
import torch
from torch import nn

from .ply_io import mkdir_p, read_ply, write_ply
from .sh_utils import RGB2SH

NUM_FEATURES = 64  # width of the per-Gaussian feature vector the reference's neural decoders read (:176)


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
        self._normals = self._features = torch.empty(0)
        self.max_radii2D = torch.empty(0)
        self.spatial_lr_scale = 0

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
        m._normals = mk(torch.zeros_like(means3D))
        m._features = mk(torch.zeros((means3D.shape[0], NUM_FEATURES), dtype=means3D.dtype))
        return m

    def create_from_pcd(self, pcd, spatial_lr_scale, device="cuda"):
        """Initialise from a point cloud with ``points``, ``colors`` in [0,1] and optional ``normals``
        (scene/gaussian_model.py:150-183): DC colour from the point colour, isotropic scale = distance scale of the
        three nearest neighbours, identity rotation, opacity 0.1.  Needs the GPU (the neighbour search runs there)."""
        from .simple_knn._C import distCUDA2
        self.spatial_lr_scale = spatial_lr_scale
        dev = torch.device(device)
        points = torch.tensor(np.asarray(pcd.points)).float().to(dev)
        n = points.shape[0]
        features = torch.zeros((n, 3, (self.max_sh_degree + 1) ** 2), dtype=torch.float32, device=dev)
        features[:, :3, 0] = RGB2SH(torch.tensor(np.asarray(pcd.colors)).float().to(dev))
        dist2 = torch.clamp_min(distCUDA2(points), 0.0000001)
        scales = torch.log(torch.sqrt(dist2))[..., None].repeat(1, 3)
        rots = torch.zeros((n, 4), device=dev)
        rots[:, 0] = 1
        opacities = inverse_sigmoid(0.1 * torch.ones((n, 1), dtype=torch.float32, device=dev))
        normals = getattr(pcd, "normals", None)
        normals = torch.zeros_like(points) if normals is None else torch.tensor(np.asarray(normals)).float().to(dev)
        mk = lambda t: nn.Parameter(t.contiguous().requires_grad_(True))
        self._xyz = mk(points)
        self._normals = mk(normals)
        self._features_dc = mk(features[:, :, 0:1].transpose(1, 2))
        self._features_rest = mk(features[:, :, 1:].transpose(1, 2))
        self._features = mk(torch.zeros((n, NUM_FEATURES), dtype=torch.float32, device=dev))
        self._scaling = mk(scales)
        self._rotation = mk(rots)
        self._opacity = mk(opacities)
        self.max_radii2D = torch.zeros((n,), device=dev)
        return self

    def construct_list_of_attributes(self):
        """Column names of a saved model, in file order (scene/gaussian_model.py:289-309)."""
        names = ["x", "y", "z", "nx", "ny", "nz"]
        names += ["f_dc_%d" % i for i in range(self._features_dc.shape[1] * self._features_dc.shape[2])]
        names += ["f_rest_%d" % i for i in range(self._features_rest.shape[1] * self._features_rest.shape[2])]
        names += ["features_%d" % i for i in range(self._features.shape[1])]
        names.append("opacity")
        names += ["scale_%d" % i for i in range(self._scaling.shape[1])]
        names += ["rot_%d" % i for i in range(self._rotation.shape[1])]
        return names

    def save_ply(self, path):
        """Raw (pre-activation) parameters as one float row per Gaussian (scene/gaussian_model.py:311-329); SH
        coefficients are stored channel-major, i.e. transposed from the [P, coefficient, channel] tensors."""
        mkdir_p(os.path.dirname(path))
        host = lambda t: t.detach().cpu().numpy()
        columns = np.concatenate((
            host(self._xyz), host(self._normals),
            host(self._features_dc.detach().transpose(1, 2).flatten(start_dim=1).contiguous()),
            host(self._features_rest.detach().transpose(1, 2).flatten(start_dim=1).contiguous()),
            host(self._features), host(self._opacity), host(self._scaling), host(self._rotation)), axis=1)
        names = self.construct_list_of_attributes()
        rows = np.empty(columns.shape[0], dtype=[(name, "f4") for name in names])
        rows.view(np.float32).reshape(columns.shape[0], len(names))[:] = columns
        write_ply(path, rows)

    def load_ply(self, path, device="cuda"):
        """Inverse of save_ply (scene/gaussian_model.py:336-392); a file without normals or feature vectors loads
        with zeros there.  The active SH degree becomes the maximum degree."""
        v = read_ply(path)
        names = v.dtype.names
        n = v.shape[0]
        by_index = lambda prefix: sorted((c for c in names if c.startswith(prefix)), key=lambda c: int(c.split("_")[-1]))
        stack = lambda cols: np.stack([np.asarray(v[c], dtype=np.float32) for c in cols], axis=1) if cols \
            else np.zeros((n, 0), np.float32)
        xyz = stack(["x", "y", "z"])
        normals = stack(["nx", "ny", "nz"]) if "nx" in names else np.zeros_like(xyz)
        features_dc = stack(["f_dc_0", "f_dc_1", "f_dc_2"]).reshape(n, 3, 1)
        rest_names = by_index("f_rest_")
        if len(rest_names) != 3 * (self.max_sh_degree + 1) ** 2 - 3:
            raise AssertionError("%s holds %d higher-band SH columns, the model's degree needs %d"
                                 % (path, len(rest_names), 3 * (self.max_sh_degree + 1) ** 2 - 3))
        features_rest = stack(rest_names).reshape(n, 3, (self.max_sh_degree + 1) ** 2 - 1)
        feature_names = by_index("features_")
        features = stack(feature_names) if feature_names else np.zeros((n, NUM_FEATURES), np.float32)
        dev = torch.device(device)
        mk = lambda a: nn.Parameter(torch.tensor(a, dtype=torch.float, device=dev).contiguous().requires_grad_(True))
        self._xyz = mk(xyz)
        self._normals = mk(normals)
        self._features_dc = nn.Parameter(torch.tensor(features_dc, dtype=torch.float, device=dev)
                                         .transpose(1, 2).contiguous().requires_grad_(True))
        self._features_rest = nn.Parameter(torch.tensor(features_rest, dtype=torch.float, device=dev)
                                           .transpose(1, 2).contiguous().requires_grad_(True))
        self._features = mk(features)
        self._opacity = mk(stack(["opacity"]))
        self._scaling = mk(stack(by_index("scale_")))
        self._rotation = mk(stack(by_index("rot")))
        self.active_sh_degree = self.max_sh_degree
        return self

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
