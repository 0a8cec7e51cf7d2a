"""Drop-in for the reference's ``rasterizer2`` package (submodules/rasterizer2/rasterizer2/__init__.py): the
nearest-centre index map the fork's neural renderer reads (gaussian_renderer/__init__.py:15,122-186).
Names, argument orders and return values are the reference's; the work is done by ``libngs_rasterizer.so``."""
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _C


def rasterize_gaussians2(means3D, normal, means2D, feature_vector, sh, scales, rotations, raster_settings):
    return _GetIdxMap2.apply(means3D, normal, means2D, feature_vector, sh, scales, rotations, raster_settings)


class _GetIdxMap2(torch.autograd.Function):
    """rasterizer2/__init__.py:31-138."""

    @staticmethod
    def forward(ctx, means3D, normal, means2D, feature_vector, sh, scales, rotations, raster_settings):
        args = (raster_settings.bg, means3D, normal, scales, rotations, raster_settings.viewmatrix,
                raster_settings.projmatrix, raster_settings.tanfovx, raster_settings.tanfovy,
                raster_settings.image_height, raster_settings.image_width, feature_vector, sh,
                raster_settings.campos, raster_settings.debug)
        # the entry count is kept in ctx by the reference but never read: do not synchronise for it
        num_rendered, idxmap, colmap, depthmap, featuremap = _C.get_idxmap2(*args, count_entries=False)
        ctx.raster_settings = raster_settings
        ctx.num_rendered = num_rendered
        ctx.save_for_backward(means3D, normal, scales, rotations, feature_vector, sh, idxmap)
        ctx.mark_non_differentiable(idxmap)
        return idxmap, colmap, depthmap, featuremap

    @staticmethod
    def backward(ctx, _a, grad_out_color, _b, grad_featuremap):
        raster_settings = ctx.raster_settings
        means3D, normal, scales, rotations, feature_vector, sh, idxmap = ctx.saved_tensors
        # only the feature vector receives a gradient (rasterizer2/__init__.py:109-137); the others get zeros
        grad_feature_vector = _C.get_idxmap_backward2(
            means3D, raster_settings.image_height, raster_settings.image_width, raster_settings.campos, idxmap,
            grad_featuremap.contiguous(), raster_settings.debug, viewmatrix=raster_settings.viewmatrix,
            projmatrix=raster_settings.projmatrix)[0]
        zeros = lambda t: torch.zeros_like(t) if t is not None else None
        return (zeros(means3D), zeros(normal), torch.zeros_like(means3D), grad_feature_vector, zeros(sh),
                zeros(scales), zeros(rotations), None)


class GaussianRasterizationSettings2(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


class GaussianRasterizer2(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def forward(self, means3D, normal, means2D, feature_vector, shs=None, scales=None, rotations=None):
        return rasterize_gaussians2(means3D, normal, means2D, feature_vector, shs, scales, rotations,
                                    self.raster_settings)
