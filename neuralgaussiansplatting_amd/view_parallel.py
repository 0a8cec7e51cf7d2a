"""View-sharded data parallelism for the rasterizer hot path.

The reference is single-GPU, one view per step (train.py:76-78,
utils/general_utils.py:133); this module adds the multi-GPU form: Gaussian
parameters are replicated, each rank (one process per GPU) rasterizes its own
camera view forward and backward, and the per-Gaussian gradients of all ranks
are summed by ONE all-reduce per step (RCCL over xGMI with the ``nccl``
backend; ``gloo`` works on CPU tensors for tests).

To make that a single collective without a packing copy, the backward pass
writes its five parameter gradients straight into slices of one flat bucket:

    [ dL_dmeans3D 3P | dL_dsh 3MP | dL_dopacity P | dL_dscales 3P | dL_drotations 4P ]
"""
import torch
import torch.distributed as dist

from .diff_gaussian_rasterization import _C

BUCKET_FIELDS = ("dL_dmeans3D", "dL_dsh", "dL_dopacity", "dL_dscales", "dL_drotations")


def bucket_layout(P, M):
    """-> ({field: (offset, length)}, total) in float32 elements."""
    sizes = (3 * P, 3 * M * P, P, 3 * P, 4 * P)
    lay, off = {}, 0
    for name, n in zip(BUCKET_FIELDS, sizes):
        lay[name] = (off, n)
        off += n
    return lay, off


class GradientBucket:
    """One flat float32 buffer holding every parameter gradient of a step."""

    def __init__(self, P, M, device):
        self.P, self.M = P, M
        self.layout, total = bucket_layout(P, M)
        self.flat = torch.empty(total, dtype=torch.float32, device=device)
        self.views = {k: self.flat[o:o + n] for k, (o, n) in self.layout.items()}

    def shaped(self):
        P, M = self.P, self.M
        shapes = dict(dL_dmeans3D=(P, 3), dL_dsh=(P, M, 3), dL_dopacity=(P, 1), dL_dscales=(P, 3),
                      dL_drotations=(P, 4))
        return {k: v.view(shapes[k]) for k, v in self.views.items()}

    def all_reduce(self, average=False, group=None, async_op=False):
        """Sum (or mean) the bucket over all ranks: the step's only collective."""
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
            return None
        work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        if average and not async_op:
            self.flat.div_(dist.get_world_size(group))
        return work


def shard_views(num_views, rank, world_size):
    """Indices of the views rank ``rank`` renders (round-robin)."""
    return list(range(rank, num_views, world_size))


def forward_backward(scene, cam, bg, dL_dpix_fn, bucket, tanfovx, tanfovy, scale_modifier=1.0, debug=False):
    """One pass of the hot path for one view on this rank's GPU: rasterize,
    obtain dL/dpixel from ``dL_dpix_fn(color)``, run the backward pass with the
    parameter gradients landing in ``bucket``.  Returns (color, radii, aux grads)."""
    e = torch.empty(0)
    H, W = int(cam.image_height), int(cam.image_width)
    R, color, radii, geom, binning, img = _C.rasterize_gaussians(
        bg, scene.means3D, e, scene.opacities, scene.scales, scene.rotations, scale_modifier, e,
        cam.world_view_transform, cam.full_proj_transform, tanfovx, tanfovy, H, W, scene.shs, scene.sh_degree,
        cam.camera_center, False, debug)
    dL_dpix = dL_dpix_fn(color)
    res = _C.rasterize_gaussians_backward(
        bg, scene.means3D, radii, e, scene.scales, scene.rotations, scale_modifier, e, cam.world_view_transform,
        cam.full_proj_transform, tanfovx, tanfovy, dL_dpix, scene.shs, scene.sh_degree, cam.camera_center, geom, R,
        binning, img, debug, out=bucket.views)
    return color, radii, dict(num_rendered=R, dL_dmeans2D=res[0], dL_dcolors=res[1], dL_dcov3D=res[4])
