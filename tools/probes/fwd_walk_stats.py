"""Utilisation counters of the forward compositing kernel's walk (library built with -DNGS_FWD_STATS)."""
import ctypes, sys, os
sys.path.insert(0, os.getcwd())
import torch
from neuralgaussiansplatting_amd import synthetic, view_parallel
from neuralgaussiansplatting_amd.diff_gaussian_rasterization import _lib as _C
cfg = sys.argv[1] if len(sys.argv) > 1 else "c4"
dev = torch.device("cuda:0")
if cfg in synthetic.CLUSTERED_CONFIGS:
    P, W, H, D, M, mean, ext = synthetic.CLUSTERED_CONFIGS[cfg]
    scene = synthetic.make_clustered_scene(P, D, mean, 1.0, ext, 1234, M).to(dev)
else:
    P, W, H, D, M, ls, ext = synthetic.CONFIGS[cfg]
    scene = synthetic.make_scene(P, D, ls, ext, 1234, M).to(dev)
cam = synthetic.make_view(W, H, index=0, count=8, device=dev)
tanx, tany = synthetic.tan_half_fov(cam)
dL = (torch.randn(3, H, W, generator=torch.Generator().manual_seed(1234)) / float(W * H)).to(dev)
bucket = view_parallel.GradientBucket(P, M, dev)
L = _C.lib()
out = (ctypes.c_ulonglong * 8)()
for _ in range(2):
    o = view_parallel.forward_backward(scene, cam, torch.zeros(3, device=dev), lambda c: dL, bucket, tanx, tany)
    torch.cuda.synchronize()
    L.ngs_debug_fwd_stats(out)
v = list(out)
R = o[2]["num_rendered"]
print(cfg, "instances", R)
print("wave-batches", v[0], "(instances x 4 / 64 = %d)" % (R * 4 // 64))
print("survivors of the first test", v[1], "per wave-batch %.1f" % (v[1] / v[0]))
print("walks", v[2], "survivors per walk %.1f" % (v[1] / max(v[2], 1)))
print("survivors with a quadrant to visit", v[3], "%.3f of the first test's" % (v[3] / v[1]))
print("list entries", v[4], "per such survivor %.2f of 4" % (v[4] / v[3]))
print("wave-steps", v[5], "per walk %.1f; row fill %.3f" % (v[5] / max(v[2], 1), v[4] / 4 / v[5]))
print("lane-visits that blended", v[6], "per wave-step %.1f of 64" % (v[6] / v[5]))
print("row-visits in which a lane blended", v[7], "%.3f of the list entries" % (v[7] / v[4]))
