#!/usr/bin/env python3
"""Utilisation counters of render_bwd's walk, from a diagnostic build of the library:

    make -C neuralgaussiansplatting_amd/csrc BUILD=build_stats OUT=../lib/variants/libngs_stats.so EXTRA=-DNGS_BWD_STATS=1 ../lib/variants/libngs_stats.so
    NGS_RASTERIZER_LIB=neuralgaussiansplatting_amd/lib/variants/libngs_stats.so python tools/probes/bwd_stats.py --config c4
"""
import argparse
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from neuralgaussiansplatting_amd import synthetic, view_parallel  # noqa: E402
from neuralgaussiansplatting_amd.diff_gaussian_rasterization import _lib as _C  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="c4")
ap.add_argument("--log-scale", type=float, nargs=2, default=None)
args = ap.parse_args()
dev = torch.device("cuda:0")
P, W, H, D, M, ls, ext = synthetic.CONFIGS[args.config]
if args.log_scale is not None:
    ls = tuple(args.log_scale)
scene = synthetic.make_scene(P, D, ls, ext, 1234, M).to(dev)
cam = synthetic.make_view(W, H, index=0, count=8, device=dev)
tanx, tany = synthetic.tan_half_fov(cam)
bg = torch.zeros(3, device=dev)
dL = (torch.randn(3, H, W, generator=torch.Generator().manual_seed(1234)) / float(W * H)).to(dev)
bucket = view_parallel.GradientBucket(P, M, dev)
lib = _C.lib()
out = (ctypes.c_ulonglong * 8)()
lib.ngs_debug_bwd_stats(out)
color, radii, aux = view_parallel.forward_backward(scene, cam, bg, lambda c: dL, bucket, tanx, tany)
torch.cuda.synchronize()
assert lib.ngs_debug_bwd_stats(out) == 0
ws, rs, live, early, wb, longest, total, staged = [int(v) for v in out]
print("instances R              %d" % aux["num_rendered"])
print("staged splats            %d  (%.2f of R)" % (staged, staged / aux["num_rendered"]))
print("wave-batches             %d" % wb)
print("wave-steps               %d  (%.1f per wave-batch)" % (ws, ws / max(wb, 1)))
print("  left at the early exit %d  (%.3f)" % (early, early / max(ws, 1)))
print("row-steps with an entry  %d  (%.3f of 4 x wave-steps)" % (rs, rs / max(4 * ws, 1)))
print("live lane-steps          %d  (%.3f of 16 x row-steps, %.3f of 64 x wave-steps)" % (live, live / max(16 * rs, 1), live / max(64 * ws, 1)))
print("longest wave per batch   sum %d vs mean %.0f  (balance %.3f)" % (longest, total / 4.0, total / 4.0 / max(longest, 1)))
