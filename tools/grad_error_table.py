#!/usr/bin/env python3
"""Largest deviation of every gradient tensor of the HIP backward pass from the CPU checker at full-size
configurations, as a markdown table (for the parity section of the design notes).  Test infrastructure: it drives
the same helpers as tests/test_parity_gpu.py.

    python tools/grad_error_table.py c4 c3 > gpurun_out/grad_errors.md
"""
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_parity_gpu as tp  # noqa: E402
from neuralgaussiansplatting_amd import diff_gaussian_rasterization as dgr, synthetic  # noqa: E402
from oracle import oracle  # noqa: E402

oracle.build()
oracle.set_threads(min(16, os.cpu_count() or 1))
print("| config | tensor | largest element | largest deviation | deviation / largest element | same, loss normalised by "
      "the pixel count |")
print("|---|---|---|---|---|---|")
for cfg in sys.argv[1:] or ["c4"]:
    sc, W, H = synthetic.make_config(cfg, seed=1234)
    cam = synthetic.make_view(W, H, 0)
    c = dict(scene=sc, cam=cam, W=W, H=H, D=sc.sh_degree, bg=torch.zeros(3), tanfovx=math.tan(cam.FoVx * 0.5),
             tanfovy=math.tan(cam.FoVy * 0.5))
    dL = torch.randn(3, H, W, generator=torch.Generator().manual_seed(8))
    o = tp.oracle_forward(oracle, c)
    h = tp.run_hip(dgr, c, dL_dpix=dL)
    og = tp.oracle_backward(oracle, c, o, dL)
    for name, ref in og.items():
        got = h["grads"][name].reshape(ref.shape)
        big, err = float(np.abs(ref).max()), float(np.abs(got - ref).max())
        print("| %s | %s | %.3g | %.3g | %.1e | %.1e |" % (cfg, name, big, err, err / big if big else 0.0, err / (W * H)),
              flush=True)
