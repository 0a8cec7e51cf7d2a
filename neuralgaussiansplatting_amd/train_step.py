"""One training iteration around the rasterizer, the counterpart of the reference's loop body (train.py:67-128):
learning-rate schedule, SH degree schedule, render, L1 + D-SSIM loss, backward, (multi-GPU: gradient exchange),
densification statistics, clone / split / prune and opacity reset on the reference's schedule, Adam.
``GaussianTrainer.step`` is the iteration for a fixed set of Gaussians (what the benchmarks time);
``GaussianTrainer.train_iteration`` wraps it with everything that changes their number or resets them.

What differs from the reference's way of running the same arithmetic:

* the SH coefficients live in ONE [P,16,3] tensor (the reference keeps ``_features_dc`` and ``_features_rest``
  apart and concatenates them every iteration, gaussian_model.py:130-133); the two learning rates are applied by
  position inside the row;
* loss value and dL/dimage come from one fused call (``_C.l1_ssim_loss``) instead of five depthwise convolutions
  and their autograd graph;
* the activations (sigmoid / exp / normalise, gaussian_model.py:36-43), their backward and Adam are one pass
  (``_C.adam_step`` / ``_C.adam_step_activated``): no autograd graph exists at all, the rasterizer's C ABI is called
  directly and its gradients wrt the activated values feed the optimiser kernels;
* on one GPU (``fused=True``, the default there) even those optimiser kernels disappear: the rasterizer reads the
  RAW parameters (``_C.rasterize_gaussians_raw``) and its per-Gaussian backward kernel applies the activations'
  backward and Adam to all five parameter groups while the gradients are still in registers
  (``_C.rasterize_gaussians_backward_train``); no gradient and no activated copy is written to memory.  Both ways
  produce bit-identical parameters.
"""
import math

import torch

from . import view_parallel
from .diff_gaussian_rasterization import _lib as _C


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
    iterations = 30_000
    percent_dense = 0.01
    densification_interval = 100
    opacity_reset_interval = 3000
    densify_from_iter = 500
    densify_until_iter = 15_000
    densify_grad_threshold = 0.0002


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

    With a process group, every rank steps its own ``views_per_rank`` views and the gradients are averaged over all
    ranks' views before Adam, so all ranks hold identical parameters; a view's collectives travel while the next
    view is rendered.  The sums of the densification statistics ride in the same collectives (every rank holds
    their totals after every step); the largest screen radius is this rank's views only until
    ``sync_densification_stats()`` merges it before densifying."""

    _ROWS = ("xyz", "features", "opacity_raw", "scaling_raw", "rotation_raw")

    def __init__(self, scene, device, opt=None, spatial_lr_scale=1.0, factored_exchange=True,
                 densify_stats=True, fused=True, views_per_rank=1):
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
        # activated copies: kept in step with the raw values by the optimiser kernel, or (fused steps, which never
        # read them) recomputed when somebody asks
        self._activate()
        self.state = {}
        for name in ("xyz", "features", "opacity_raw", "scaling_raw", "rotation_raw"):
            p = getattr(self, name)
            self.state[name] = (torch.zeros_like(p), torch.zeros_like(p))
        self.iteration = 0
        # torch.optim.Adam counts steps per parameter, and the reference's parameters miss the step of an iteration
        # in which they were replaced (densification: all of them; opacity reset: the opacities) -- train.py:113-128
        self.adam_steps = dict.fromkeys(("xyz", "features", "opacity_raw", "scaling_raw", "rotation_raw"), 0)
        self.max_sh_degree = max(int(round(math.sqrt(self.M))) - 1, self.sh_degree)
        self.extras = {}  # further per-Gaussian rows (normals, feature vectors ...) that follow clone / split / prune
        world = torch.distributed.get_world_size() if (torch.distributed.is_available() and
                                                       torch.distributed.is_initialized()) else 1
        self.world = world
        self.views_per_rank = int(views_per_rank)
        self.factored_exchange = factored_exchange
        self.densify_stats = densify_stats
        # one GPU, full SH rows: the whole optimiser step runs inside the backward pass
        self.fused = bool(fused) and world == 1 and self.views_per_rank == 1 and self.M == 16 and dev.type == "cuda"
        self._workspace = None
        self._dL_dpix = None
        self._resized()

    @classmethod
    def from_model(cls, model, device, **kw):
        """Take over a ``GaussianModel``'s raw parameters (fresh from ``create_from_pcd`` or ``load_ply``), the way
        ``training_setup`` hands them to the optimiser (gaussian_model.py:252-283).  Normals and the per-Gaussian
        feature vectors ride along in ``extras``."""
        from .synthetic import Scene
        with torch.no_grad():
            scene = Scene(model.get_xyz, model.get_features, model.get_opacity, model.get_scaling, model.get_rotation,
                          model.active_sh_degree)
            tr = cls(scene, device, spatial_lr_scale=model.spatial_lr_scale, **kw)
            f = lambda t: t.detach().to(tr.xyz.device, torch.float32).contiguous().clone()
            tr.opacity_raw, tr.scaling_raw, tr.rotation_raw = f(model._opacity), f(model._scaling), f(model._rotation)
            tr._activate()
            tr.max_sh_degree = model.max_sh_degree
            tr.extras = dict(normals=f(model._normals), features64=f(model._features))
        return tr

    def to_model(self):
        """The current parameters as a ``GaussianModel`` (for ``save_ply`` or ``render()``)."""
        from torch import nn
        from .gaussian_model import GaussianModel
        m = GaussianModel(self.max_sh_degree)
        m.active_sh_degree = self.sh_degree
        m.spatial_lr_scale = self.spatial_lr_scale
        mk = lambda t: nn.Parameter(t.detach().clone().contiguous().requires_grad_(True))
        m._xyz, m._opacity, m._scaling, m._rotation = mk(self.xyz), mk(self.opacity_raw), mk(self.scaling_raw), \
            mk(self.rotation_raw)
        m._features_dc, m._features_rest = mk(self.features[:, :1]), mk(self.features[:, 1:])
        zeros = lambda w: torch.zeros((self.P, w), device=self.xyz.device)
        m._normals = mk(self.extras.get("normals", zeros(3)))
        from .gaussian_model import NUM_FEATURES
        m._features = mk(self.extras.get("features64", zeros(NUM_FEATURES)))
        if self.densify_stats:
            m.max_radii2D = self.max_radii2D.float()
        return m

    def capture(self):
        """Everything a resumed run needs, as one dict of tensors and numbers -- the counterpart of
        ``GaussianModel.capture()`` (gaussian_model.py:75-91) for ``torch.save((trainer.capture(), iteration))``
        (train.py:130-132)."""
        out = dict(iteration=self.iteration, sh_degree=self.sh_degree, max_sh_degree=self.max_sh_degree,
                   spatial_lr_scale=self.spatial_lr_scale, adam_steps=dict(self.adam_steps),
                   params={n: getattr(self, n).clone() for n in self._ROWS},
                   exp_avg={n: self.state[n][0].clone() for n in self._ROWS},
                   exp_avg_sq={n: self.state[n][1].clone() for n in self._ROWS},
                   extras={n: t.clone() for n, t in self.extras.items()})
        if self.densify_stats:
            out["stats"] = dict(xyz_gradient_accum=self.xyz_gradient_accum.clone(), denom=self.denom.clone(),
                                max_radii2D=self.max_radii2D.clone())
        return out

    def restore(self, captured):
        """Inverse of ``capture()`` (gaussian_model.py:93-113); the trainer continues bit for bit."""
        dev = self.xyz.device
        f = lambda t: t.detach().to(dev).contiguous().clone()
        for n in self._ROWS:
            setattr(self, n, f(captured["params"][n]))
            self.state[n] = (f(captured["exp_avg"][n]), f(captured["exp_avg_sq"][n]))
        self.extras = {n: f(t) for n, t in captured["extras"].items()}
        self.iteration, self.sh_degree = int(captured["iteration"]), int(captured["sh_degree"])
        self.max_sh_degree, self.spatial_lr_scale = int(captured["max_sh_degree"]), captured["spatial_lr_scale"]
        self.adam_steps = dict(captured["adam_steps"])
        self.M = self.features.shape[1]
        self._activate()
        self._resized()
        if self.densify_stats and "stats" in captured:
            st = captured["stats"]
            self.xyz_gradient_accum, self.denom, self.max_radii2D = (f(st["xyz_gradient_accum"]), f(st["denom"]),
                                                                      f(st["max_radii2D"]))
        return self

    def _resized(self):
        """(Re)build everything whose size follows the number of Gaussians: gradient buffers, statistics."""
        dev = self.xyz.device
        self.P = self.xyz.shape[0]
        # several views per step (over ranks or on one rank): the statistics' sums travel with the gradients
        self.batched = self.world > 1 or self.views_per_rank > 1
        kw = dict(stats=self.densify_stats and self.batched, views_per_rank=self.views_per_rank)
        self.exchange = (view_parallel.FactoredGradientExchange(self.P, self.M, dev, **kw)
                         if self.batched and self.factored_exchange
                         else view_parallel.GradientBucket(self.P, self.M, dev, **kw))
        if self.densify_stats:  # gaussian_model.py:486-488
            self.xyz_gradient_accum = torch.zeros((self.P, 1), device=dev)
            self.denom = torch.zeros((self.P, 1), device=dev)
            self.max_radii2D = torch.zeros(self.P, dtype=torch.int32, device=dev)

    def _activate(self):
        """Activated copies of the raw parameters (gaussian_model.py:36-43).  On the GPU they come from the library,
        whose raw-parameter kernels apply the very same expressions."""
        if self.opacity_raw.is_cuda:
            self._opacities, self._scales, self._rotations = _C.activate_parameters(
                self.opacity_raw, self.scaling_raw, self.rotation_raw)
        else:
            self._opacities = torch.sigmoid(self.opacity_raw)
            self._scales = torch.exp(self.scaling_raw)
            self._rotations = torch.nn.functional.normalize(self.rotation_raw)
        self._stale = False

    def _activated(self, name):
        if self._stale:
            self._activate()
        return getattr(self, name)

    opacities = property(lambda self: self._activated("_opacities"))
    scales = property(lambda self: self._activated("_scales"))
    rotations = property(lambda self: self._activated("_rotations"))

    def oneupSHdegree(self):  # gaussian_model.py:146-148
        if self.sh_degree < self.max_sh_degree:
            self.sh_degree += 1

    def sync_densification_stats(self, group=None):
        """Totals over all ranks' views, in place (see ``view_parallel.sync_densification_stats``); the sums are
        already global, they travelled with every step's gradients."""
        if self.densify_stats:
            view_parallel.sync_densification_stats(self.xyz_gradient_accum, self.denom, self.max_radii2D, group,
                                                   sums=False)

    def scene_view(self):
        """The activated tensors in the shape ``view_parallel.forward_backward`` expects."""
        from .synthetic import Scene
        return Scene(self.xyz, self.features, self.opacities, self.scales, self.rotations, self.sh_degree)

    def xyz_lr(self, iteration):
        o = self.opt
        return expon_lr(iteration, o.position_lr_init * self.spatial_lr_scale,
                        o.position_lr_final * self.spatial_lr_scale, lr_delay_mult=o.position_lr_delay_mult,
                        max_steps=o.position_lr_max_steps)

    def step(self, cam, gt_image, bg, tanfovx, tanfovy, optimize=True, hold_opacity=False, collect_stats=True):
        """One iteration on this rank's view -- or views: with ``views_per_rank`` > 1, ``cam`` and ``gt_image`` are
        sequences of that length.  Returns (loss_terms, color, radii) of the (last) view; loss_terms stays on the
        device.  ``optimize=False`` renders and differentiates (statistics are still gathered) but moves no
        parameter; ``hold_opacity`` leaves the opacities alone while everything else takes its step."""
        if self.batched:
            return self._batched_step(cam, gt_image, bg, tanfovx, tanfovy, optimize, hold_opacity, collect_stats)
        self.iteration += 1
        it, o = self.iteration, self.opt
        stats = self.densify_stats and collect_stats
        H, W = int(cam.image_height), int(cam.image_width)
        if self._workspace is None or self._dL_dpix is None or self._dL_dpix.shape != (3, H, W):
            self._workspace = torch.empty(_C.loss_workspace_floats(W, H), dtype=torch.float32, device=self.xyz.device)
            self._dL_dpix = torch.empty((3, H, W), dtype=torch.float32, device=self.xyz.device)
        terms = {}

        def loss_grad(color):
            terms["loss"], dL = _C.l1_ssim_loss(color, gt_image, o.lambda_dssim, workspace=self._workspace,
                                                out=self._dL_dpix)
            return dL

        if self.fused and optimize:
            return self._fused_step(cam, gt_image, bg, tanfovx, tanfovy, loss_grad, terms, hold_opacity, stats)
        scene = self.scene_view()
        # On one rank the SH coefficients (four fifths of all parameters) are stepped inside the backward pass: their
        # gradient is consumed where it is produced instead of travelling through memory to a separate Adam kernel.
        if not optimize:
            color, radii, aux = view_parallel.forward_backward(scene, cam, bg, loss_grad, self.exchange, tanfovx,
                                                               tanfovy)
            if stats:
                self._add_densification_stats(aux["dL_dmeans2D"], radii)
            return terms["loss"], color, radii
        steps = self.adam_steps
        for name in steps:
            if not (hold_opacity and name == "opacity_raw"):
                steps[name] += 1
        fused_sh = self.M == 16
        sh_adam = None
        if fused_sh:  # inside the backward pass
            m, v = self.state["features"]
            sh_adam = dict(param=self.features, exp_avg=m, exp_avg_sq=v, lr_dc=o.feature_lr,
                           lr_rest=o.feature_lr / 20.0, step=steps["features"], grad_scale=1.0)
        color, radii, aux = view_parallel.forward_backward(scene, cam, bg, loss_grad, self.exchange, tanfovx, tanfovy,
                                                           sh_adam=sh_adam)
        densify = None
        if stats:
            densify = (aux["dL_dmeans2D"], radii, self.xyz_gradient_accum, self.denom, self.max_radii2D)
        self._adam_from_exchange(steps, 1.0, fused_sh, hold_opacity, densify)
        return terms["loss"], color, radii

    def _adam_from_exchange(self, steps, scale, sh_done, hold_opacity, densify):
        """The optimiser kernels on the gradients the exchange holds (``scale`` turns sums over views into means),
        run by run of Gaussians as the exchange's buffers are laid out (one run unless the exchange is chunked)."""
        it, o = self.iteration, self.opt
        names = ("opacity_raw", "scaling_raw", "rotation_raw")
        for first, count, g in self.exchange.chunk_views():
            rows = slice(first, first + count)
            m, v = self.state["xyz"]
            _C.adam_step(self.xyz[rows], g["dL_dmeans3D"], m[rows], v[rows], self.xyz_lr(it), steps["xyz"],
                         grad_scale=scale)
            if not sh_done:
                m, v = self.state["features"]
                dL_dsh = g["dL_dsh"] if "dL_dsh" in g else self.exchange.dL_dsh[rows]
                _C.adam_step(self.features[rows], dL_dsh, m[rows], v[rows], o.feature_lr, steps["features"],
                             lr_tail=o.feature_lr / 20.0, period=3 * self.M, head=3, grad_scale=scale)
            # a held opacity takes a step of size zero (its moments are about to be cleared by the reset that asked
            # for it)
            _C.adam_step_activated(
                tuple(getattr(self, n)[rows] for n in names),
                (g["dL_dopacity"], g["dL_dscales"], g["dL_drotations"]),
                tuple(self.state[n][0][rows] for n in names), tuple(self.state[n][1][rows] for n in names),
                (self._opacities[rows], self._scales[rows], self._rotations[rows]),
                (0.0 if hold_opacity else o.opacity_lr, o.scaling_lr, o.rotation_lr),
                tuple(max(steps[n], 1) for n in names), grad_scale=scale,
                densify=None if densify is None else tuple(t[rows] for t in densify))

    def _batched_step(self, cams, gts, bg, tanfovx, tanfovy, optimize, hold_opacity, collect_stats):
        """One iteration whose loss is the mean over ``views_per_rank`` views on each of ``world`` ranks.  View k's
        gradients (and statistics) leave for the other ranks as soon as its backward pass is queued and travel while
        view k+1 is rendered; only the last view's exchange is exposed."""
        V = self.views_per_rank
        if V == 1 and not isinstance(cams, (list, tuple)):
            cams, gts = [cams], [gts]
        if not isinstance(cams, (list, tuple)) or len(cams) != V or len(gts) != V:
            raise RuntimeError("this trainer steps %d views per rank and iteration" % V)
        self.iteration += 1
        o, ex = self.opt, self.exchange
        stats = self.densify_stats and collect_stats
        scene = self.scene_view()
        terms = {}
        for k, (cam, gt) in enumerate(zip(cams, gts)):
            H, W = int(cam.image_height), int(cam.image_width)
            if self._workspace is None or self._dL_dpix is None or self._dL_dpix.shape != (3, H, W):
                self._workspace = torch.empty(_C.loss_workspace_floats(W, H), dtype=torch.float32,
                                              device=self.xyz.device)
                self._dL_dpix = torch.empty((3, H, W), dtype=torch.float32, device=self.xyz.device)

            def loss_grad(color):
                terms["loss"], dL = _C.l1_ssim_loss(color, gt, o.lambda_dssim, workspace=self._workspace,
                                                    out=self._dL_dpix)
                return dL

            color, radii, aux = view_parallel.forward_backward(scene, cam, bg, loss_grad, ex, tanfovx, tanfovy, slot=k,
                                                               max_radii2D=self.max_radii2D if stats else None)
            ex.post_view(k, scene, cam, aux)
        scale = 1.0 / (self.world * V)  # the batch loss is the mean over the views of all ranks
        steps = self.adam_steps
        factored = isinstance(ex, view_parallel.FactoredGradientExchange)
        fused_sh = optimize and self.M == 16 and factored
        sh_adam = None
        if optimize:
            for name in steps:
                if not (hold_opacity and name == "opacity_raw"):
                    steps[name] += 1
        if fused_sh:  # inside the kernel that rebuilds the all-view SH gradient
            m, v = self.state["features"]
            sh_adam = dict(param=self.features, exp_avg=m, exp_avg_sq=v, lr_dc=o.feature_lr,
                           lr_rest=o.feature_lr / 20.0, step=steps["features"], grad_scale=scale)
        ex.finish(scene, sh_adam=sh_adam)
        if stats and ex.stats:  # totals over every view of every rank
            for first, count, g in ex.chunk_views():
                self.xyz_gradient_accum[first:first + count].add_(g["stat_norm"])
                self.denom[first:first + count].add_(g["stat_vis"])
        if optimize:
            self._adam_from_exchange(steps, scale, fused_sh, hold_opacity, None)
        return terms["loss"], color, radii

    def _fused_step(self, cam, gt_image, bg, tanfovx, tanfovy, loss_grad, terms, hold_opacity, stats):
        """The iteration with the optimiser inside the backward pass: raw parameters in, updated raw parameters
        out, nothing in between reaches memory but the image and its gradient."""
        o, steps = self.opt, self.adam_steps
        for name in steps:
            if not (hold_opacity and name == "opacity_raw"):
                steps[name] += 1
        e = torch.empty(0)
        H, W = int(cam.image_height), int(cam.image_width)
        R, color, radii, geom, binning, img = _C.rasterize_gaussians_raw(
            bg, self.xyz, e, self.opacity_raw, self.scaling_raw, self.rotation_raw, 1.0, cam.world_view_transform,
            cam.full_proj_transform, tanfovx, tanfovy, H, W, self.features, self.sh_degree, cam.camera_center, False,
            False)
        dL_dpix = loss_grad(color)
        lrs = dict(xyz=self.xyz_lr(self.iteration), features=o.feature_lr,
                   opacity_raw=0.0 if hold_opacity else o.opacity_lr, scaling_raw=o.scaling_lr,
                   rotation_raw=o.rotation_lr)
        names = dict(xyz="xyz", features="features", opacity="opacity_raw", scaling="scaling_raw",
                     rotation="rotation_raw")
        groups = {k: (getattr(self, n),) + self.state[n] + (lrs[n], max(steps[n], 1)) for k, n in names.items()}
        densify = (self.xyz_gradient_accum, self.denom, self.max_radii2D) if stats else None
        _C.rasterize_gaussians_backward_train(
            bg, radii, 1.0, cam.world_view_transform, cam.full_proj_transform, tanfovx, tanfovy, dL_dpix,
            self.sh_degree, cam.camera_center, geom, R, binning, img, groups, o.feature_lr / 20.0,
            grad_scale=1.0 / self.world, densify=densify)
        self._stale = True
        return terms["loss"], color, radii

    # ---- everything below changes the set of Gaussians: host-side tensor surgery as in the reference ----

    def _add_densification_stats(self, dL_dmeans2D, radii):
        """train.py:115 and gaussian_model.py:550-552 for an iteration whose optimiser step is skipped."""
        vis = radii > 0
        self.max_radii2D[vis] = torch.max(self.max_radii2D[vis], radii[vis])
        self.xyz_gradient_accum[vis] += torch.norm(dL_dmeans2D[vis, :2], dim=-1, keepdim=True)
        self.denom[vis] += 1

    def _append_rows(self, new, new_extras):
        """gaussian_model.py:444-488: new Gaussians start with zero Adam moments; the statistics restart."""
        for name in self._ROWS:
            add = new[name]
            setattr(self, name, torch.cat((getattr(self, name), add), dim=0).contiguous())
            m, v = self.state[name]
            self.state[name] = (torch.cat((m, torch.zeros_like(add)), dim=0).contiguous(),
                                torch.cat((v, torch.zeros_like(add)), dim=0).contiguous())
        for name, add in new_extras.items():
            self.extras[name] = torch.cat((self.extras[name], add), dim=0)
        self._activate()
        self._resized()

    def prune_points(self, mask):
        """Drop the Gaussians where ``mask`` is set, with their Adam moments and statistics (:409-442)."""
        keep = ~mask
        for name in self._ROWS:
            setattr(self, name, getattr(self, name)[keep].contiguous())
            m, v = self.state[name]
            self.state[name] = (m[keep].contiguous(), v[keep].contiguous())
        for name in self.extras:
            self.extras[name] = self.extras[name][keep]
        stats = (self.xyz_gradient_accum[keep], self.denom[keep], self.max_radii2D[keep]) if self.densify_stats else None
        self._activate()
        self._resized()
        if stats is not None:
            self.xyz_gradient_accum, self.denom, self.max_radii2D = (t.contiguous() for t in stats)

    def densify_and_clone(self, grads, grad_threshold, scene_extent):
        """Small Gaussians with a large view-space gradient are duplicated in place (:517-532)."""
        sel = torch.norm(grads, dim=-1) >= grad_threshold
        sel &= self.scales.max(dim=1).values <= self.opt.percent_dense * scene_extent
        self._append_rows({n: getattr(self, n)[sel] for n in self._ROWS}, {n: t[sel] for n, t in self.extras.items()})

    def densify_and_split(self, grads, grad_threshold, scene_extent, N=2, generator=None):
        """Large ones are replaced by N samples of themselves, scaled down by 0.8 N (:490-515)."""
        from .gaussian_model import quaternion_to_rotation
        padded = torch.zeros(self.P, device=self.xyz.device)
        padded[:grads.shape[0]] = grads.squeeze(-1)
        sel = padded >= grad_threshold
        sel &= self.scales.max(dim=1).values > self.opt.percent_dense * scene_extent
        scales = torch.exp(self.scaling_raw[sel])
        stds = scales.repeat(N, 1)
        samples = torch.normal(mean=torch.zeros_like(stds), std=stds, generator=generator)
        rots = quaternion_to_rotation(self.rotation_raw[sel]).repeat(N, 1, 1)
        new = dict(
            xyz=torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + self.xyz[sel].repeat(N, 1),
            scaling_raw=torch.log(scales.repeat(N, 1) / (0.8 * N)),
            rotation_raw=self.rotation_raw[sel].repeat(N, 1),
            features=self.features[sel].repeat(N, 1, 1),
            opacity_raw=self.opacity_raw[sel].repeat(N, 1))
        self._append_rows(new, {n: t[sel].repeat(N, *([1] * (t.dim() - 1))) for n, t in self.extras.items()})
        added = N * int(sel.sum())
        self.prune_points(torch.cat((sel, torch.zeros(added, dtype=torch.bool, device=sel.device))))

    def densify_and_prune(self, max_grad, min_opacity, extent, max_screen_size, generator=None):
        """gaussian_model.py:534-548.  With several ranks the statistics are merged first and ``generator`` must be
        seeded alike everywhere, so that every rank makes the same decisions and draws the same samples."""
        self.sync_densification_stats()
        grads = self.xyz_gradient_accum / self.denom
        grads[grads.isnan()] = 0.0
        self.densify_and_clone(grads, max_grad, extent)
        self.densify_and_split(grads, max_grad, extent, generator=generator)
        prune = (self.opacities < min_opacity).squeeze(-1)
        if max_screen_size:
            big_vs = self.max_radii2D > max_screen_size
            big_ws = self.scales.max(dim=1).values > 0.1 * extent
            prune = prune | big_vs | big_ws
        self.prune_points(prune)

    def reset_opacity(self):
        """Opacities above 0.01 fall back to it; the opacities' Adam moments restart (:331-334, :394-407)."""
        cap = torch.full_like(self.opacity_raw, 0.01)
        self.opacity_raw = inverse_sigmoid(torch.min(torch.sigmoid(self.opacity_raw), cap)).contiguous()
        self.state["opacity_raw"] = (torch.zeros_like(self.opacity_raw), torch.zeros_like(self.opacity_raw))
        self._activate()

    def train_iteration(self, cam, gt_image, bg, tanfovx, tanfovy, scene_extent, white_background=False, seed=0):
        """The whole loop body of train.py:69-128 for iteration ``self.iteration + 1``.

        An iteration that densifies replaces every parameter before the optimiser runs, so none takes a step; one
        that only resets the opacities steps everything else.  The last iteration takes no step either."""
        it, o = self.iteration + 1, self.opt
        if it % 1000 == 0:
            self.oneupSHdegree()
        growing = it < o.densify_until_iter
        densify_now = growing and it > o.densify_from_iter and it % o.densification_interval == 0
        reset_now = growing and (it % o.opacity_reset_interval == 0 or
                                 (white_background and it == o.densify_from_iter))
        out = self.step(cam, gt_image, bg, tanfovx, tanfovy, optimize=not densify_now and it < o.iterations,
                        hold_opacity=reset_now, collect_stats=growing)
        if densify_now:
            gen = torch.Generator(device=self.xyz.device)
            gen.manual_seed(1_000_003 * seed + it)
            size_threshold = 20 if it > o.opacity_reset_interval else None
            self.densify_and_prune(o.densify_grad_threshold, 0.005, scene_extent, size_threshold, generator=gen)
        if reset_now:
            self.reset_opacity()
        return out
