import sys, time, torch
sys.path.insert(0, ".")
from neuralgaussiansplatting_amd import synthetic
from neuralgaussiansplatting_amd.rasterizer2 import _C as r2c
dev = torch.device("cuda:0")
P, W, H, D, M, _, _ = synthetic.CONFIGS["c4"]
scene, W, H = synthetic.make_config("c4", seed=1234)
scene = scene.to(dev)
cam = synthetic.make_view(W, H, index=0, count=8, device=dev)
tanx, tany = synthetic.tan_half_fov(cam)
fvec = torch.randn(P, 64, device=dev); gfeat = torch.randn(H, W, 64, device=dev); e0 = torch.empty(0, device=dev)
o = r2c.get_idxmap2(e0, scene.means3D, e0, e0, e0, cam.world_view_transform, cam.full_proj_transform, tanx, tany, H, W, fvec, e0, cam.camera_center, False, count_entries=True)
print("entries", o[0], "pixels hit", int((o[1] >= 0).sum()), "distinct", int(torch.unique(o[1]).numel()))
for name, kw in (("gather", dict(viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform)), ("atomic", {})):
    for _ in range(3): r2c.get_idxmap_backward2(scene.means3D, H, W, cam.camera_center, o[1], gfeat, False, **kw)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): r2c.get_idxmap_backward2(scene.means3D, H, W, cam.camera_center, o[1], gfeat, False, **kw)
    torch.cuda.synchronize(); print(name, "%.3f ms" % ((time.perf_counter() - t) / 10 * 1e3))
