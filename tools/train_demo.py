"""End-to-end run of the training loop on a synthetic scene, the counterpart of the reference's `train.py` run on a
Blender-style scene (random viewpoint per iteration from a stack, train.py:75-78; evaluation PSNR, train.py:156-191).

A ground-truth scene of Gaussians is rendered from a ring of cameras; a sparse, colour-less subset of its centres is
the initial point cloud (`create_from_pcd`, scales from the nearest-neighbour search); training then runs with the
reference's schedule -- densification from iteration 500 every 100, opacity reset every 3000, SH degree +1 every 1000
-- and PSNR is reported on held-out views.  Prints one JSON object.

    python tools/train_demo.py [--iterations 3000] [--size 800] [--gt 30000] [--init 6000] [--out FILE]
"""
import argparse
import json
import math
import os
import random
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from neuralgaussiansplatting_amd import synthetic, train_step  # noqa: E402
from neuralgaussiansplatting_amd.diff_gaussian_rasterization import _C  # noqa: E402
from neuralgaussiansplatting_amd.gaussian_model import GaussianModel  # noqa: E402
from neuralgaussiansplatting_amd.ply_io import BasicPointCloud  # noqa: E402


def render_scene(scene, cam, bg):
    e = torch.empty(0)
    tanx, tany = synthetic.tan_half_fov(cam)
    out = _C.rasterize_gaussians(bg, scene.means3D, e, scene.opacities, scene.scales, scene.rotations, 1.0, e,
                                 cam.world_view_transform, cam.full_proj_transform, tanx, tany, int(cam.image_height),
                                 int(cam.image_width), scene.shs, scene.sh_degree, cam.camera_center, False, False)
    return out[1]


def psnr(a, b):
    return float(10.0 * torch.log10(1.0 / ((a - b) ** 2).mean()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iterations", type=int, default=3000)
    ap.add_argument("--size", type=int, default=800)
    ap.add_argument("--gt", type=int, default=30000, help="Gaussians of the ground-truth scene")
    ap.add_argument("--init", type=int, default=6000, help="points of the initial cloud")
    ap.add_argument("--views", type=int, default=40)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    random.seed(0)
    truth = synthetic.make_scene(args.gt, 3, (-3.9, -2.9), (1.0, 1.0, 1.0), seed=77, sh_coeffs=16).to(dev)
    bg = torch.zeros(3, device=dev)
    W = H = args.size
    cams = [synthetic.make_view(W, H, i, args.views, device=dev) for i in range(args.views)]
    images = [render_scene(truth, c, bg).clone() for c in cams]
    held_out = list(range(0, args.views, 8))
    train_ids = [i for i in range(args.views) if i not in held_out]
    # initial cloud: a subset of the true centres, jittered, grey
    g = np.random.default_rng(5)
    pick = g.choice(args.gt, size=args.init, replace=False)
    pts = truth.means3D.cpu().numpy()[pick] + g.normal(0.0, 0.01, (args.init, 3)).astype(np.float32)
    pcd = BasicPointCloud(points=pts.astype(np.float32), colors=np.full((args.init, 3), 0.5, np.float32),
                          normals=np.zeros((args.init, 3), np.float32))
    extent = 4.0 * 1.1  # radius of the camera ring, scene/dataset_readers.py:48-66
    model = GaussianModel(3).create_from_pcd(pcd, spatial_lr_scale=extent, device=dev)
    tr = train_step.GaussianTrainer.from_model(model, dev)
    tr.opt.iterations = args.iterations

    def evaluate():
        view = tr.scene_view()
        return sum(psnr(render_scene(view, cams[i], bg), images[i]) for i in held_out) / len(held_out)

    log = [dict(iteration=0, gaussians=tr.P, psnr=round(evaluate(), 3))]
    stack = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loop_s = 0.0
    for it in range(1, args.iterations + 1):
        if not stack:
            stack = train_ids.copy()
        i = stack.pop(random.randint(0, len(stack) - 1))
        cam = cams[i]
        tanx, tany = synthetic.tan_half_fov(cam)
        tr.train_iteration(cam, images[i], bg, tanx, tany, scene_extent=extent)
        if it % 500 == 0 or it == args.iterations:
            torch.cuda.synchronize()
            loop_s += time.perf_counter() - t0
            log.append(dict(iteration=it, gaussians=tr.P, sh_degree=tr.sh_degree, psnr=round(evaluate(), 3),
                            iters_per_s=round(it / loop_s, 1)))
            print(json.dumps(log[-1]), file=sys.stderr, flush=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
    res = dict(what="synthetic end-to-end training run (point cloud -> densification schedule -> held-out PSNR)",
               image=[W, H], truth_gaussians=args.gt, initial_points=args.init, train_views=len(train_ids),
               held_out_views=len(held_out), iterations=args.iterations, adam_steps=tr.adam_steps, log=log,
               device=torch.cuda.get_device_name(0))
    line = json.dumps(res)
    print(line)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            f.write(line + "\n")
    assert math.isfinite(log[-1]["psnr"])


if __name__ == "__main__":
    main()
