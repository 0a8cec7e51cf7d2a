#!/usr/bin/env python3
"""Counters of the instrumented backward kernel (a -DNGS_BWD_STATS build named by NGS_RASTERIZER_LIB): steps walked,
list lengths, barrier-coupled steps, live lanes.  One forward+backward of a configuration."""
import ctypes
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
L.ngs_debug_bwd_stats(out, 1)
color, radii, aux = view_parallel.forward_backward(scene, cam, torch.zeros(3, device=dev), lambda c: dL, bucket, tanx, tany)
torch.cuda.synchronize()
L.ngs_debug_bwd_stats(out, 1)
v = list(out)
T = ((W + 15) // 16) * ((H + 15) // 16)
print("R", aux["num_rendered"], "tiles", T)
print("wave steps walked        %12d  per tile %.1f" % (v[0], v[0] / T))
print("row steps with a splat   %12d  -> row balance %.3f (sum n_q / 4 / steps)" % (v[1], v[1] / 4 / max(v[0], 1)))
print("candidates tested        %12d  per tile %.1f (tested again after a chunk did not fit: %d)" % (v[2], v[2] / T, v[5]))
print("survivors parked         %12d  per tile %.1f; quadrants per survivor %.2f" % (v[3], v[3] / T, v[1] / max(v[3], 1)))
print("walks started            %12d  steps per walk %.1f" % (v[4], v[0] / max(v[4], 1)))
print("live lanes               %12d  per row step with a splat %.2f of 16" % (v[6], v[6] / max(v[1], 1)))
print("2x2 blocks with a live px %11d  per row step with a splat %.2f of 4; live pixels per such block %.2f of 4" % (v[7], v[7] / max(v[1], 1), v[6] / max(v[7], 1)))
