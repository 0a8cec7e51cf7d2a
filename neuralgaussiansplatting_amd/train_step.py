"""One training iteration around the rasterizer, the counterpart of the reference's loop body (train.py:67-128)
for a fixed set of Gaussians: learning-rate schedule, render, L1 + D-SSIM loss, backward, (multi-GPU: gradient
exchange), Adam.  Densification, pruning and opacity reset change the number of Gaussians and are not part of it;
the statistics they consume (gaussian_model.py:550-552, train.py:115) are maintained.

What differs from the reference's way of running the same arithmetic:

* the SH coefficients live in ONE [P,16,3] tensor (the reference keeps ``_features_dc`` and ``_features_rest``
  apart and concatenates them every iteration, gaussian_model.py:130-133); the two learning rates are applied by
  position inside the row;
* loss value and dL/dimage come from one fused call (``_C.l1_ssim_loss``) instead of five depthwise convolutions
  and their autograd graph;
* the activations (sigmoid / exp / normalise, gaussian_model.py:36-43), their backward and Adam are one pass
  (``_C.adam_step`` / ``_C.adam_step_activated``): no autograd graph exists at all, the rasterizer's C ABI is called
  directly and its gradients wrt the activated values feed the optimiser kernels.
"""
import math

import torch

from . import view_parallel
from .diff_gaussian_rasterization import _C


class OptimizationParams:
    """Defaults of arguments/__init__.py:70-90."""
    position_lr_init = 0.00016
    position_lr_final = 0.0000016
    position_lr_delay_mult = 0.01
    position_lr_max_steps = 30_000
    feature_lr = 0.0025
    opacity_lr = 0.05
    scaling_lr = 0.005
    rotation_lr = 0.001
    lambda_dssim = 0.2


def expon_lr(step, lr_init, lr_final, lr_delay_steps=0, lr_delay_mult=1.0, max_steps=1000000):
    """Log-linear learning-rate decay with an optional eased-in start (utils/general_utils.py:29-61)."""
    if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
        return 0.0
    if lr_delay_steps > 0:
        delay_rate = lr_delay_mult + (1 - lr_delay_mult) * math.sin(
            0.5 * math.pi * min(max(step / lr_delay_steps, 0.0), 1.0))
    else:
        delay_rate = 1.0
    t = min(max(step / max_steps, 0.0), 1.0)
    return delay_rate * math.exp(math.log(lr_init) * (1 - t) + math.log(lr_final) * t)


def inverse_sigmoid(x):
    return torch.log(x / (1 - x))  # utils/general_utils.py:18-19


class GaussianTrainer:
    """Raw parameters, Adam state and the activated copies the rasterizer reads, all resident on one GPU.

    With a process group, every rank steps its own view and the gradients are averaged over the ranks' views before
    Adam, so all ranks hold identical parameters.  The densification statistics are this rank's views only:
    ``sync_densification_stats()`` merges them over the ranks before densifying."""

    def __init__(self, scene, device, opt=None, spatial_lr_scale=1.0, factored_exchange=True,
                 densify_stats=True):
        self.opt = opt or OptimizationParams()
        self.spatial_lr_scale = spatial_lr_scale
        dev = torch.device(device)
        self.P, self.M, self.sh_degree = scene.means3D.shape[0], scene.shs.shape[1], scene.sh_degree
        f = lambda t: t.detach().to(dev, torch.float32).contiguous().clone()
        # raw parameters; positions and SH coefficients have no activation, so the rasterizer reads them directly
        self.xyz = f(scene.means3D)
        self.features = f(scene.shs)
        self.opacity_raw = inverse_sigmoid(f(scene.opacities))
        self.scaling_raw = torch.log(f(scene.scales))
        self.rotation_raw = f(scene.rotations)
        # activated copies, kept in step with the raw values by the optimiser kernel
        self.opacities = torch.sigmoid(self.opacity_raw)
        self.scales = torch.exp(self.scaling_raw)
        self.rotations = torch.nn.functional.normalize(self.rotation_raw)
        self.state = {}
        for name in ("xyz", "features", "opacity_raw", "scaling_raw", "rotation_raw"):
            p = getattr(self, name)
            self.state[name] = (torch.zeros_like(p), torch.zeros_like(p))
        self.iteration = 0
        world = torch.distributed.get_world_size() if (torch.distributed.is_available() and
                                                       torch.distributed.is_initialized()) else 1
        self.world = world
        self.exchange = (view_parallel.FactoredGradientExchange(self.P, self.M, dev) if world > 1 and factored_exchange
                         else view_parallel.GradientBucket(self.P, self.M, dev))
        self.densify = None
        if densify_stats:
            self.xyz_gradient_accum = torch.zeros((self.P, 1), device=dev)
            self.denom = torch.zeros((self.P, 1), device=dev)
            self.max_radii2D = torch.zeros(self.P, dtype=torch.int32, device=dev)
        self.densify_stats = densify_stats
        self._workspace = None
        self._dL_dpix = None

    def sync_densification_stats(self, group=None):
        """Totals over all ranks' views, in place (see ``view_parallel.sync_densification_stats``)."""
        if self.densify_stats:
            view_parallel.sync_densification_stats(self.xyz_gradient_accum, self.denom, self.max_radii2D, group)

    def scene_view(self):
        """The activated tensors in the shape ``view_parallel.forward_backward`` expects."""
        from .synthetic import Scene
        return Scene(self.xyz, self.features, self.opacities, self.scales, self.rotations, self.sh_degree)

    def xyz_lr(self, iteration):
        o = self.opt
        return expon_lr(iteration, o.position_lr_init * self.spatial_lr_scale,
                        o.position_lr_final * self.spatial_lr_scale, lr_delay_mult=o.position_lr_delay_mult,
                        max_steps=o.position_lr_max_steps)

    def step(self, cam, gt_image, bg, tanfovx, tanfovy):
        """One iteration on this rank's view.  Returns (loss_terms, color, radii); loss_terms stays on the device."""
        self.iteration += 1
        it, o = self.iteration, self.opt
        H, W = int(cam.image_height), int(cam.image_width)
        if self._workspace is None or self._dL_dpix is None or self._dL_dpix.shape != (3, H, W):
            self._workspace = torch.empty(_C.loss_workspace_floats(W, H), dtype=torch.float32, device=self.xyz.device)
            self._dL_dpix = torch.empty((3, H, W), dtype=torch.float32, device=self.xyz.device)
        terms = {}

        def loss_grad(color):
            terms["loss"], dL = _C.l1_ssim_loss(color, gt_image, o.lambda_dssim, workspace=self._workspace,
                                                out=self._dL_dpix)
            return dL

        scene = self.scene_view()
        # On one rank the SH coefficients (four fifths of all parameters) are stepped inside the backward pass: their
        # gradient is consumed where it is produced instead of travelling through memory to a separate Adam kernel.
        scale = 1.0 / self.world  # the batch loss is the mean over the views of all ranks
        factored = isinstance(self.exchange, view_parallel.FactoredGradientExchange)
        fused_sh = self.M == 16 and (self.world == 1 or factored)
        sh_adam = None
        if fused_sh:
            m, v = self.state["features"]
            sh_adam = dict(param=self.features, exp_avg=m, exp_avg_sq=v, lr_dc=o.feature_lr,
                           lr_rest=o.feature_lr / 20.0, step=it, grad_scale=scale)
        # one rank: inside the backward pass; several ranks: inside the kernel that rebuilds the all-view SH gradient
        color, radii, aux = view_parallel.forward_backward(scene, cam, bg, loss_grad, self.exchange, tanfovx, tanfovy,
                                                           sh_adam=sh_adam if self.world == 1 else None)
        self.exchange.reduce(scene, cam, aux, sh_adam=sh_adam if (self.world > 1 and fused_sh) else None)
        g = self.exchange.shaped()
        m, v = self.state["xyz"]
        _C.adam_step(self.xyz, g["dL_dmeans3D"], m, v, self.xyz_lr(it), it, grad_scale=scale)
        if not fused_sh:
            m, v = self.state["features"]
            _C.adam_step(self.features, g["dL_dsh"], m, v, o.feature_lr, it, lr_tail=o.feature_lr / 20.0,
                         period=3 * self.M, head=3, grad_scale=scale)
        names = ("opacity_raw", "scaling_raw", "rotation_raw")
        densify = None
        if self.densify_stats:
            densify = (aux["dL_dmeans2D"], radii, self.xyz_gradient_accum, self.denom, self.max_radii2D)
        _C.adam_step_activated(
            tuple(getattr(self, n) for n in names),
            (g["dL_dopacity"], g["dL_dscales"], g["dL_drotations"]),
            tuple(self.state[n][0] for n in names), tuple(self.state[n][1] for n in names),
            (self.opacities, self.scales, self.rotations), (o.opacity_lr, o.scaling_lr, o.rotation_lr), it,
            grad_scale=scale, densify=densify)
        return terms["loss"], color, radii
