"""Times the view-factored SH gradient rebuild (and its fused Adam form) on one GPU: P Gaussians, V views.
    python tools/sh_views_bench.py [P] [V]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuralgaussiansplatting_amd.diff_gaussian_rasterization import _C  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000
V = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
means = torch.randn(P, 3, device=dev, generator=g)
campos = torch.randn(V, 3, device=dev, generator=g) * 4
colors = torch.randn(V, 3 * P, device=dev, generator=g)
out = torch.empty(P, 16, 3, device=dev)
sh, m, v = torch.randn(P, 16, 3, device=dev, generator=g), torch.zeros(P, 16, 3, device=dev), torch.zeros(P, 16, 3, device=dev)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


t1 = timed(lambda: _C.sh_gradient_from_views(means, campos, colors, 3, 16, out=out))
t2 = timed(lambda: _C.sh_gradient_from_views(means, campos, colors, 3, 16, sh_adam=dict(
    param=sh, exp_avg=m, exp_avg_sq=v, lr_dc=0.0025, lr_rest=0.000125, step=5, grad_scale=1.0 / V)))
gb1 = (P * (12 + 192) + V * P * 12) / 1e9
gb2 = (P * (12 + 6 * 192) + V * P * 12) / 1e9
print("P=%d V=%d: gradient %.3f ms (%.0f GB/s), fused Adam %.3f ms (%.0f GB/s)" % (P, V, t1, gb1 / t1 * 1e3, t2, gb2 / t2 * 1e3))
