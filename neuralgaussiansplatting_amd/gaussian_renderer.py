"""Caller-side counterpart of the reference's ``render()``
(gaussian_renderer/__init__.py:19-101): builds the raster settings from a
camera and a Gaussian model, chooses the SH / covariance code paths from the
pipeline flags, calls the rasterizer and packages the result dict.

``pc`` is any object with the reference model's read interface
(scene/gaussian_model.py:113-144): ``get_xyz, get_opacity, get_scaling,
get_rotation, get_features, get_covariance(scaling_modifier),
active_sh_degree, max_sh_degree``.  ``pipe`` carries ``debug``,
``compute_cov3D_python`` and ``convert_SHs_python``
(arguments/__init__.py:64-69).
"""
import math

import torch

from .diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
from .sh_utils import eval_sh


class PipelineParams:
    """Defaults of the reference's pipeline flag group (arguments/__init__.py:64-69)."""

    def __init__(self, convert_SHs_python=False, compute_cov3D_python=False, debug=False):
        self.convert_SHs_python = convert_SHs_python
        self.compute_cov3D_python = compute_cov3D_python
        self.debug = debug


def render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None):
    """Render the scene.  ``bg_color`` must live on the GPU.  Returns
    ``{"render", "viewspace_points", "visibility_filter", "radii"}``."""
    xyz = pc.get_xyz
    # zero tensor whose gradient receives dL/d(screen-space mean); densification reads it
    screenspace_points = torch.zeros_like(xyz, dtype=xyz.dtype, requires_grad=True, device=xyz.device) + 0
    try:
        screenspace_points.retain_grad()
    except Exception:
        pass

    settings = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height),
        image_width=int(viewpoint_camera.image_width),
        tanfovx=math.tan(viewpoint_camera.FoVx * 0.5),
        tanfovy=math.tan(viewpoint_camera.FoVy * 0.5),
        bg=bg_color,
        scale_modifier=scaling_modifier,
        viewmatrix=viewpoint_camera.world_view_transform,
        projmatrix=viewpoint_camera.full_proj_transform,
        sh_degree=pc.active_sh_degree,
        campos=viewpoint_camera.camera_center,
        prefiltered=False,
        debug=pipe.debug,
    )
    rasterizer = GaussianRasterizer(raster_settings=settings)

    scales = rotations = cov3D_precomp = None
    if pipe.compute_cov3D_python:
        cov3D_precomp = pc.get_covariance(scaling_modifier)
    else:
        scales, rotations = pc.get_scaling, pc.get_rotation

    shs = colors_precomp = None
    if override_color is not None:
        colors_precomp = override_color
    elif pipe.convert_SHs_python:
        feats = pc.get_features
        shs_view = feats.transpose(1, 2).view(-1, 3, (pc.max_sh_degree + 1) ** 2)
        dir_pp = xyz - viewpoint_camera.camera_center.repeat(feats.shape[0], 1)
        dir_pp = dir_pp / dir_pp.norm(dim=1, keepdim=True)
        colors_precomp = torch.clamp_min(eval_sh(pc.active_sh_degree, shs_view, dir_pp) + 0.5, 0.0)
    else:
        shs = pc.get_features

    rendered_image, radii = rasterizer(means3D=xyz, means2D=screenspace_points, shs=shs,
                                       colors_precomp=colors_precomp, opacities=pc.get_opacity, scales=scales,
                                       rotations=rotations, cov3D_precomp=cov3D_precomp)
    return {"render": rendered_image, "viewspace_points": screenspace_points, "visibility_filter": radii > 0,
            "radii": radii}
