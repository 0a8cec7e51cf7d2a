"""Host-side behaviour of the drop-in package that needs no GPU: public
surface, argument validation (reference __init__.py:187-195), error paths."""
import inspect
import re

import pytest
import torch

from neuralgaussiansplatting_amd import diff_gaussian_rasterization as dgr
from neuralgaussiansplatting_amd.gaussian_model import GaussianModel, covariance_from_scaling_rotation
from neuralgaussiansplatting_amd.sh_utils import eval_sh


def settings():
    return dgr.GaussianRasterizationSettings(
        image_height=16, image_width=16, tanfovx=0.3, tanfovy=0.3, bg=torch.zeros(3), scale_modifier=1.0,
        viewmatrix=torch.eye(4), projmatrix=torch.eye(4), sh_degree=0, campos=torch.zeros(3), prefiltered=False,
        debug=False)


def test_public_surface_matches_reference_names_and_order():
    assert dgr.GaussianRasterizationSettings._fields == (
        "image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix", "projmatrix",
        "sh_degree", "campos", "prefiltered", "debug")
    fwd = inspect.signature(dgr.GaussianRasterizer.forward)
    assert list(fwd.parameters) == ["self", "means3D", "means2D", "opacities", "shs", "colors_precomp", "scales",
                                    "rotations", "cov3D_precomp"]
    assert list(inspect.signature(dgr.rasterize_gaussians).parameters) == [
        "means3D", "means2D", "sh", "colors_precomp", "opacities", "scales", "rotations", "cov3Ds_precomp",
        "raster_settings"]
    assert issubclass(dgr._RasterizeGaussians, torch.autograd.Function)
    for name in ("rasterize_gaussians", "rasterize_gaussians_backward", "mark_visible"):
        assert callable(getattr(dgr._C, name))
    # _C is a compiled extension like the reference's (ext.cpp:15-19); pybind11 records the argument names of
    # rasterize_points.h:18-67 in the first line of each docstring
    def names(fn):
        head = fn.__doc__.splitlines()[0]
        return re.findall(r"(\w+): ", head[head.index("("):head.rindex(") ->")])
    assert type(dgr._C.rasterize_gaussians).__name__ == "builtin_function_or_method"
    assert dgr._lib.rasterize_gaussians is dgr._C.rasterize_gaussians  # re-exported, not wrapped
    assert names(dgr._C.rasterize_gaussians) == [
        "background", "means3D", "colors", "opacity", "scales", "rotations", "scale_modifier", "cov3D_precomp",
        "viewmatrix", "projmatrix", "tan_fovx", "tan_fovy", "image_height", "image_width", "sh", "degree", "campos",
        "prefiltered", "debug"]
    assert names(dgr._C.rasterize_gaussians_backward) == [
        "background", "means3D", "radii", "colors", "scales", "rotations", "scale_modifier", "cov3D_precomp",
        "viewmatrix", "projmatrix", "tan_fovx", "tan_fovy", "dL_dout_color", "sh", "degree", "campos", "geomBuffer", "R",
        "binningBuffer", "imageBuffer", "debug"]
    assert names(dgr._C.mark_visible) == ["means3D", "viewmatrix", "projmatrix"]


def test_exactly_one_colour_source():
    r = dgr.GaussianRasterizer(settings())
    m, o = torch.zeros(4, 3), torch.zeros(4, 1)
    with pytest.raises(Exception, match="SHs or precomputed colors"):
        r(m, m, o, scales=torch.ones(4, 3), rotations=torch.ones(4, 4))
    with pytest.raises(Exception, match="SHs or precomputed colors"):
        r(m, m, o, shs=torch.zeros(4, 1, 3), colors_precomp=torch.zeros(4, 3), scales=torch.ones(4, 3),
          rotations=torch.ones(4, 4))


def test_exactly_one_covariance_source():
    r = dgr.GaussianRasterizer(settings())
    m, o, sh = torch.zeros(4, 3), torch.zeros(4, 1), torch.zeros(4, 1, 3)
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(m, m, o, shs=sh)
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(m, m, o, shs=sh, scales=torch.ones(4, 3))  # half a pair
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(m, m, o, shs=sh, scales=torch.ones(4, 3), rotations=torch.ones(4, 4), cov3D_precomp=torch.ones(4, 6))


def test_cpu_tensors_are_refused_not_silently_computed():
    r = dgr.GaussianRasterizer(settings())
    m = torch.zeros(4, 3)
    with pytest.raises(RuntimeError, match="GPU only"):
        r(m, m, torch.zeros(4, 1), shs=torch.zeros(4, 1, 3), scales=torch.ones(4, 3), rotations=torch.ones(4, 4))
    with pytest.raises(RuntimeError, match="GPU only"):
        r.markVisible(m)
    with pytest.raises(RuntimeError, match="means3D must have dimensions"):
        dgr._lib.rasterize_gaussians(torch.zeros(3), torch.zeros(4, 2), *([torch.empty(0)] * 4), 1.0, torch.empty(0),
                                   torch.eye(4), torch.eye(4), 0.3, 0.3, 16, 16, torch.empty(0), 0, torch.zeros(3),
                                   False, False)


def test_missing_library_is_a_loud_error(monkeypatch):
    monkeypatch.setattr(dgr._lib, "_lib", None)
    monkeypatch.setattr(dgr._lib, "LIB_PATH", "/nonexistent/libngs_rasterizer.so")
    with pytest.raises(RuntimeError, match="not built"):
        dgr._lib.lib()


def test_gaussian_model_activations_and_covariance():
    g = torch.Generator().manual_seed(0)
    q = torch.randn(10, 4, generator=g)
    s = torch.rand(10, 3, generator=g) + 0.1
    m = GaussianModel.from_activated(torch.randn(10, 3, generator=g), torch.randn(10, 16, 3, generator=g),
                                     torch.rand(10, 1, generator=g) * 0.8 + 0.1, s, q, sh_degree=3)
    assert m.get_features.shape == (10, 16, 3)
    torch.testing.assert_close(m.get_scaling, s)
    torch.testing.assert_close(m.get_rotation.norm(dim=1), torch.ones(10))
    cov = covariance_from_scaling_rotation(s, 1.0, q)
    full = torch.zeros(10, 3, 3)
    idx = [(0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)]
    for k, (i, j) in enumerate(idx):
        full[:, i, j] = cov[:, k]
        full[:, j, i] = cov[:, k]
    ev = torch.linalg.eigvalsh(full)
    torch.testing.assert_close(ev, torch.sort(s * s, dim=1).values, rtol=1e-4, atol=1e-6)


def test_eval_sh_shapes_and_dc():
    sh = torch.zeros(5, 3, 16)
    sh[..., 0] = 2.0
    d = torch.nn.functional.normalize(torch.randn(5, 3), dim=1)
    for deg in range(4):
        torch.testing.assert_close(eval_sh(deg, sh, d), torch.full((5, 3), 2.0 * 0.28209479177387814))
    with pytest.raises(ValueError):
        eval_sh(4, sh, d)
