#!/usr/bin/env python3
"""Per-stage times of the forward pass from activated parameters and from the optimiser's raw parameters on one scene."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from neuralgaussiansplatting_amd import synthetic
from neuralgaussiansplatting_amd.diff_gaussian_rasterization import _C, _lib

cfg = sys.argv[1] if len(sys.argv) > 1 else "c4"
dev = torch.device("cuda:0")
P, W, H, D, M, ls, ext = synthetic.CONFIGS[cfg]
sc = synthetic.make_scene(P, D, ls, ext, 1234, M).to(dev)
cam = synthetic.make_view(W, H, index=0, count=8, device=dev)
tanx, tany = synthetic.tan_half_fov(cam)
bg = torch.zeros(3, device=dev)
e = torch.empty(0)
op_raw = torch.log(sc.opacities / (1 - sc.opacities)).contiguous()
sc_raw = torch.log(sc.scales).contiguous()
rot_raw = (sc.rotations * 1.7).contiguous()


def run(raw):
    _lib.profile_enable(True)
    _lib.profile_read(reset=True)
    for _ in range(30):
        if raw:
            _lib.rasterize_gaussians_raw(bg, sc.means3D, e, op_raw, sc_raw, rot_raw, 1.0, cam.world_view_transform,
                                         cam.full_proj_transform, tanx, tany, H, W, sc.shs, D, cam.camera_center, False, False)
        else:
            _C.rasterize_gaussians(bg, sc.means3D, e, sc.opacities, sc.scales, sc.rotations, 1.0, e, cam.world_view_transform,
                                   cam.full_proj_transform, tanx, tany, H, W, sc.shs, D, cam.camera_center, False, False)
    torch.cuda.synchronize()
    st = _lib.profile_read(reset=True)
    _lib.profile_enable(False)
    return {k: tot / cnt for k, (tot, cnt) in st.items() if cnt}


for raw in (False, True, False, True):
    print("raw" if raw else "activated", {k: round(v, 4) for k, v in run(raw).items()})
