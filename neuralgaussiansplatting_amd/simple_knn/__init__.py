"""Drop-in for the reference's ``simple_knn`` package (submodules/simple-knn): ``from simple_knn._C import distCUDA2``
(scene/gaussian_model.py:20) resolves to ``neuralgaussiansplatting_amd.simple_knn._C.distCUDA2``."""
from . import _C  # noqa: F401
