import ctypes, sys, os
sys.path.insert(0, os.getcwd())
import torch
from neuralgaussiansplatting_amd import synthetic, view_parallel
from neuralgaussiansplatting_amd.diff_gaussian_rasterization import _lib as _C
cfg = sys.argv[1] if len(sys.argv) > 1 else "c4"
dev = torch.device("cuda:0")
P, W, H, D, M, ls, ext = synthetic.CONFIGS[cfg]
scene = synthetic.make_scene(P, D, ls, ext, 1234, M).to(dev)
cam = synthetic.make_view(W, H, index=0, count=8, device=dev)
tanx, tany = synthetic.tan_half_fov(cam)
dL = (torch.randn(3, H, W, generator=torch.Generator().manual_seed(1234)) / float(W * H)).to(dev)
bucket = view_parallel.GradientBucket(P, M, dev)
L = _C.lib()
out = (ctypes.c_ulonglong * 8)()
view_parallel.forward_backward(scene, cam, torch.zeros(3, device=dev), lambda c: dL, bucket, tanx, tany)
torch.cuda.synchronize()
L.ngs_debug_bwd_stats(out)
view_parallel.forward_backward(scene, cam, torch.zeros(3, device=dev), lambda c: dL, bucket, tanx, tany)
torch.cuda.synchronize()
L.ngs_debug_bwd_stats(out)
v = list(out)
print("wave-steps", v[0], "row-steps with entry", v[1], "row balance %.3f" % (v[1] / 4 / v[0]))
print("live lane-steps", v[2], "per row-step %.2f of 16" % (v[2] / v[1]))
print("early exits", v[3], "%.3f of wave-steps" % (v[3] / v[0]))
print("wave-batches", v[4], "steps per wave-batch %.1f" % (v[0] / v[4]))
print("shared steps", v[5], "%.3f of wave-steps" % (v[5] / v[0]))
print("staged splats", v[7])
