"""Times the fused L1 + D-SSIM loss (value and gradient) alone on a 1080p image: the kernels are short enough that
a profile of this script shows them without the rest of the training iteration."""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from neuralgaussiansplatting_amd.diff_gaussian_rasterization import _C  # noqa: E402

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
dev = torch.device("cuda:0")
img = torch.rand(3, H, W, device=dev)
gt = torch.rand(3, H, W, device=dev)
ws = torch.empty(_C.loss_workspace_floats(W, H), device=dev)
out = torch.empty_like(img)
for _ in range(5):
    _C.l1_ssim_loss(img, gt, 0.2, workspace=ws, out=out)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 50
for _ in range(n):
    _C.l1_ssim_loss(img, gt, 0.2, workspace=ws, out=out)
torch.cuda.synchronize()
print("l1_ssim_loss %dx%d: %.1f us per call" % (W, H, (time.perf_counter() - t0) / n * 1e6))
