"""GPU tests of the training-step kernels (through the C ABI): the fused image loss against a fixture produced by
the reference's loss functions under autograd, the fused Adam / activation kernels against torch.optim.Adam and
torch autograd, and whole iterations against the same iteration assembled from PyTorch operators."""
import math
import os

import numpy as np
import pytest
import torch

from helpers import free_port
from neuralgaussiansplatting_amd import loss_utils, synthetic, train_step
from neuralgaussiansplatting_amd.cameras import make_camera

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DEV = "cuda:0"


@pytest.fixture(scope="module")
def C():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from neuralgaussiansplatting_amd.diff_gaussian_rasterization import _lib as _C
    _C.lib()
    return _C


def test_fused_loss_matches_reference_fixture(C):
    z = np.load(os.path.join(GOLDEN, "train_loss.npz"))
    img, gt = torch.tensor(z["img"], device=DEV), torch.tensor(z["gt"], device=DEV)
    terms, dL = C.l1_ssim_loss(img, gt, float(z["lambda_dssim"]))
    terms = terms.cpu().numpy()
    assert abs(terms[0] - float(z["l1"])) < 1e-6
    assert abs(terms[1] - float(z["ssim"])) < 1e-6
    assert abs(terms[2] - float(z["loss"])) < 1e-6
    g = z["grad"]
    # gradient entries are O(1/(3*H*W)); 1e-4 of the largest one is ~1e-9 absolute
    assert float(np.abs(dL.cpu().numpy() - g).max()) <= 1e-4 * float(np.abs(g).max())


@pytest.mark.parametrize("H,W", [(16, 16), (7, 5), (64, 48), (33, 130), (75, 41)])
def test_fused_loss_matches_torch_formulation(C, H, W):
    g = torch.Generator().manual_seed(H * 1000 + W)
    img = torch.rand(3, H, W, generator=g).to(DEV).requires_grad_(True)
    gt = torch.rand(3, H, W, generator=g).to(DEV)
    loss = loss_utils.training_loss(img, gt, 0.2)
    loss.backward()
    terms, dL = C.l1_ssim_loss(img.detach(), gt, 0.2)
    assert abs(float(terms[2]) - loss.item()) < 2e-6
    ref = img.grad
    assert float((dL - ref).abs().max()) <= 2e-4 * float(ref.abs().max())


# NGS_RANDOM_CONFIGS=N widens the sweep for a soak run
@pytest.mark.parametrize("seed", range(int(os.environ.get("NGS_RANDOM_CONFIGS", "6"))))
def test_fused_loss_random_sizes(C, seed):
    rng = np.random.RandomState(9000 + seed)
    H, W = int(rng.randint(1, 200)), int(rng.randint(1, 300))
    lam = float(rng.choice([0.2, 0.0, 1.0, 0.5]))
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(3, H, W, generator=g).to(DEV).requires_grad_(True)
    gt = (torch.rand(3, H, W, generator=g) if seed % 3 else img.detach().cpu() + 0.01 * torch.randn(3, H, W, generator=g)).to(DEV)
    loss = loss_utils.training_loss(img, gt, lam)
    loss.backward()
    terms, dL = C.l1_ssim_loss(img.detach(), gt, lam)
    assert abs(float(terms[2]) - loss.item()) < 2e-6
    ref = img.grad
    assert float((dL - ref).abs().max()) <= 2e-4 * float(ref.abs().max()) + 1e-9


def _torch_adam(params, lrs, grads_per_step, eps=1e-15):
    opt = torch.optim.Adam([{"params": [p], "lr": lr} for p, lr in zip(params, lrs)], lr=0.0, eps=eps)
    for grads in grads_per_step:
        for p, g in zip(params, grads):
            p.grad = g.clone()
        opt.step()
    return opt


def test_adam_step_matches_torch_adam_with_two_rates(C):
    gen = torch.Generator().manual_seed(3)
    P, M = 501, 16
    p0 = torch.randn(P, M, 3, generator=gen)
    grads = [torch.randn(P, M, 3, generator=gen) * (10.0 ** -k) for k in range(3)]
    grads[1][::7] = 0.0  # Gaussians unseen in a step have exactly zero gradient
    dc = torch.nn.Parameter(p0[:, :1].clone().to(DEV))
    rest = torch.nn.Parameter(p0[:, 1:].clone().to(DEV))
    _torch_adam([dc, rest], [0.0025, 0.0025 / 20.0],
                [(g[:, :1].contiguous().to(DEV), g[:, 1:].contiguous().to(DEV)) for g in grads])
    p = p0.clone().to(DEV)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for t, g in enumerate(grads, start=1):
        C.adam_step(p, g.to(DEV), m, v, 0.0025, t, lr_tail=0.0025 / 20.0, period=3 * M, head=3)
    want = torch.cat([dc.detach(), rest.detach()], dim=1)
    assert float((p - want).abs().max()) <= 2e-7
    # odd length (tail path) and a gradient scale
    n = 1003
    q0, g0 = torch.randn(n, generator=gen), torch.randn(n, generator=gen)
    q = torch.nn.Parameter(q0.clone().to(DEV))
    _torch_adam([q], [0.01], [(0.25 * g0.to(DEV),)])
    q2, m2, v2 = q0.clone().to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    C.adam_step(q2, g0.to(DEV), m2, v2, 0.01, 1, grad_scale=0.25)
    assert float((q2 - q.detach()).abs().max()) <= 2e-7
    with pytest.raises(RuntimeError):
        C.adam_step(q2, g0.to(DEV), m2, v2, 0.01, 0)  # steps count from 1


def test_adam_step_activated_matches_autograd_chain(C):
    gen = torch.Generator().manual_seed(5)
    P = 777
    raw0 = [torch.randn(P, 1, generator=gen) * 2, torch.randn(P, 3, generator=gen) - 4.0,
            torch.randn(P, 4, generator=gen)]
    act_grads = [[torch.randn(P, w, generator=gen) * s for w in (1, 3, 4)] for s in (1.0, 1e-3, 10.0)]
    lrs = (0.05, 0.005, 0.001)
    # reference formulation: parameters -> activations (gaussian_model.py:36-43) -> autograd -> torch.optim.Adam
    params = [torch.nn.Parameter(r.clone().to(DEV)) for r in raw0]
    opt = torch.optim.Adam([{"params": [p], "lr": lr} for p, lr in zip(params, lrs)], lr=0.0, eps=1e-15)
    acts = (torch.sigmoid, torch.exp, torch.nn.functional.normalize)
    for grads in act_grads:
        opt.zero_grad(set_to_none=True)
        sum((f(p) * g.to(DEV)).sum() for f, p, g in zip(acts, params, grads)).backward()
        opt.step()
    raw = [r.clone().to(DEV) for r in raw0]
    m = [torch.zeros_like(r) for r in raw]
    v = [torch.zeros_like(r) for r in raw]
    activated = [torch.empty_like(r) for r in raw]
    radii = (torch.arange(P) % 3).to(torch.int32).to(DEV)  # a third of the Gaussians culled
    g2d = torch.randn(P, 3, generator=gen).to(DEV)
    accum, denom = torch.ones(P, 1, device=DEV), torch.zeros(P, 1, device=DEV)
    maxr = torch.ones(P, dtype=torch.int32, device=DEV)
    for t, grads in enumerate(act_grads, start=1):
        C.adam_step_activated(raw, [g.to(DEV) for g in grads], m, v, activated, lrs, t,
                              densify=(g2d, radii, accum, denom, maxr) if t == 1 else None)
    for got, p, f in zip(raw, params, acts):
        assert float((got - p.detach()).abs().max()) <= 5e-6 * max(1.0, float(p.detach().abs().max()))
    for got, p, f in zip(activated, params, acts):
        assert float((got - f(p.detach())).abs().max()) <= 1e-6
    vis = radii > 0
    want_accum = torch.ones(P, 1, device=DEV)
    want_accum[vis] += torch.norm(g2d[vis, :2], dim=-1, keepdim=True)  # gaussian_model.py:551
    assert torch.allclose(accum, want_accum, atol=1e-6)
    assert torch.equal(denom, vis.float().reshape(P, 1))
    assert torch.equal(maxr, torch.maximum(torch.ones_like(radii), radii))


def test_fused_sh_adam_from_views_equals_gradient_then_adam(C):
    """Multi-view form of the fused step: rebuilding the all-view SH gradient and stepping Adam in one kernel must
    give exactly what the two separate kernels give."""
    gen = torch.Generator().manual_seed(12)
    P, V, D = 1300, 3, 3
    means = torch.randn(P, 3, generator=gen).to(DEV)
    campos = (torch.randn(V, 3, generator=gen) * 4).to(DEV)
    colors = torch.randn(V, P * 3, generator=gen).to(DEV)
    colors[1, : 3 * 200] = 0.0
    sh0 = torch.randn(P, 16, 3, generator=gen).to(DEV)
    m0 = (torch.randn(P, 16, 3, generator=gen) * 0.1).to(DEV)
    v0 = (torch.rand(P, 16, 3, generator=gen) * 0.01).to(DEV)
    kw = dict(lr_dc=0.0025, lr_rest=0.0025 / 20.0, step=7, grad_scale=1.0 / V)
    grad = C.sh_gradient_from_views(means, campos, colors, D, 16)
    p1, m1, v1 = sh0.clone(), m0.clone(), v0.clone()
    C.adam_step(p1, grad, m1, v1, kw["lr_dc"], kw["step"], lr_tail=kw["lr_rest"], period=48, head=3,
                grad_scale=kw["grad_scale"])
    p2, m2, v2 = sh0.clone(), m0.clone(), v0.clone()
    assert C.sh_gradient_from_views(means, campos, colors, D, 16,
                                    sh_adam=dict(param=p2, exp_avg=m2, exp_avg_sq=v2, **kw)) is None
    torch.cuda.synchronize()
    assert torch.equal(p1, p2) and torch.equal(m1, m2) and torch.equal(v1, v2)
    assert not torch.equal(p2, sh0)


@pytest.mark.parametrize("config", ["small", "c3"])
def test_training_iterations_match_pytorch_assembly(C, config):
    """Three iterations of GaussianTrainer against the same iterations assembled the reference's way: separate
    f_dc / f_rest parameters concatenated per iteration, activations and loss as PyTorch operators under autograd
    around render(), torch.optim.Adam with the reference's parameter groups.  ``c3`` is BASELINE configuration 3
    (300 k Gaussians, 800x800, SH degree 3, training loop) at full size."""
    from neuralgaussiansplatting_amd.gaussian_model import GaussianModel
    from neuralgaussiansplatting_amd.gaussian_renderer import PipelineParams, render
    if config == "c3":
        sc, W, H = synthetic.make_config("c3")
        P, D = sc.means3D.shape[0], 3
    else:
        P, W, H, D = 4000, 160, 96, 3
        sc = synthetic.make_scene(P, D, (-4.0, -3.0), (1.3, 1.3, 1.3), seed=3, sh_coeffs=16)
    cams = [make_camera(W, H, synthetic.FOVX, e, device=DEV) for e in ((0.7, 0.5, -4.0), (-2.0, 0.4, -3.4), (1.5, -0.3, -3.6))]
    gts = [torch.rand(3, H, W, generator=torch.Generator().manual_seed(40 + i)).to(DEV) for i in range(3)]
    bg = torch.zeros(3, device=DEV)
    tr = train_step.GaussianTrainer(sc, DEV)
    o = tr.opt

    model = GaussianModel.from_activated(sc.means3D, sc.shs, sc.opacities, sc.scales, sc.rotations, D, device=DEV)
    xyz, f_dc, f_rest, op, scl, rot = (model._xyz, model._features_dc, model._features_rest, model._opacity,
                                       model._scaling, model._rotation)
    groups = [(xyz, o.position_lr_init), (f_dc, o.feature_lr), (f_rest, o.feature_lr / 20.0), (op, o.opacity_lr),
              (scl, o.scaling_lr), (rot, o.rotation_lr)]
    opt = torch.optim.Adam([{"params": [p], "lr": lr} for p, lr in groups], lr=0.0, eps=1e-15)
    pipe = PipelineParams()
    for it in range(1, 4):
        cam, gt = cams[it - 1], gts[it - 1]
        tanx, tany = math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5)
        terms, color, radii = tr.step(cam, gt, bg, tanx, tany)
        opt.param_groups[0]["lr"] = tr.xyz_lr(it)  # train.py:69
        pkg = render(cam, model, pipe, bg)
        loss = loss_utils.training_loss(pkg["render"], gt, o.lambda_dssim)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        assert abs(float(terms[2]) - loss.item()) < 1e-5
        # A splat may sit on the other side of the 1/255 alpha threshold in a few pixels of a large image: in the
        # first iteration (identical parameters, exp differing by an ulp) at most two pixels per 100 000 may differ
        # by more than 1e-4; later the two sides have taken slightly different Adam steps (see below: a rounding-
        # level gradient may step either way by a whole learning rate, for up to 0.5 % of the elements), which
        # moves more pixels across -- up to five in 10 000 -- but none by more than one faint splat's
        # contribution, and the images stay equal on average to float32 rounding.
        diff = (color - pkg["render"].detach()).abs().amax(dim=0)
        few = 0 if config == "small" else int((2e-5 if it == 1 else 5e-4) * W * H)
        assert int((diff >= 1e-4).sum()) <= few
        assert float(diff.max()) < 2.0 / 255.0 and float(diff.mean()) < 1e-6
    got = dict(xyz=tr.xyz, f=tr.features, op=tr.opacity_raw, scl=tr.scaling_raw, rot=tr.rotation_raw)
    want = dict(xyz=xyz.detach(), f=torch.cat((f_dc, f_rest), dim=1).detach(), op=op.detach(), scl=scl.detach(),
                rot=rot.detach())
    lr_of = dict(xyz=o.position_lr_init, f=o.feature_lr, op=o.opacity_lr, scl=o.scaling_lr, rot=o.rotation_lr)
    for k in got:
        diff = (got[k] - want[k]).abs()
        # Adam's first steps move every element by about lr * sign(gradient): an element whose gradient is at
        # rounding level (float atomics order differs between two backward passes) may step either way, so a
        # handful of elements may differ by up to a few lr; everything else must agree closely.
        close = diff <= 1e-3 * lr_of[k]
        assert float(close.float().mean()) > 0.995, k
        assert float(diff.max()) <= 6.5 * lr_of[k], k
    assert int((tr.denom > 0).sum()) > 0 and int(tr.max_radii2D.max()) > 0


def test_training_fits_a_rendered_target(C):
    """End to end: the image of a scene is the target, a copy of that scene with scrambled colours and dimmed
    opacities is trained on it from three views; the loss has to fall substantially and monotonically enough."""
    from neuralgaussiansplatting_amd.diff_gaussian_rasterization import _lib as dC
    P, W, H, D = 3000, 128, 96, 3
    truth = synthetic.make_scene(P, D, (-3.6, -2.8), (1.3, 1.0, 1.3), seed=12, sh_coeffs=16).to(DEV)
    cams = [make_camera(W, H, synthetic.FOVX, e, device=DEV) for e in ((0.7, 0.5, -4.0), (-1.5, 0.4, -3.6), (1.4, -0.3, -3.7))]
    bg = torch.zeros(3, device=DEV)
    e = torch.empty(0)
    targets = []
    for cam in cams:
        tanx, tany = math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5)
        out = dC.rasterize_gaussians(bg, truth.means3D, e, truth.opacities, truth.scales, truth.rotations, 1.0, e,
                                     cam.world_view_transform, cam.full_proj_transform, tanx, tany, H, W, truth.shs, D,
                                     cam.camera_center, False, False)
        targets.append(out[1].clone())
    start = synthetic.make_scene(P, D, (-3.6, -2.8), (1.3, 1.0, 1.3), seed=12, sh_coeffs=16)
    g = torch.Generator().manual_seed(5)
    start.shs[:, 0, :] = torch.rand(P, 3, generator=g) * 2.0 - 1.0
    start.shs[:, 1:, :] = 0.0
    start.opacities.mul_(0.5)
    tr = train_step.GaussianTrainer(start, DEV)
    losses = []
    for it in range(90):
        cam, gt = cams[it % 3], targets[it % 3]
        terms, _, _ = tr.step(cam, gt, bg, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5))
        losses.append(float(terms[2]))
    first, last = sum(losses[:3]) / 3, sum(losses[-3:]) / 3
    assert last < 0.5 * first, (first, last)
    assert all(torch.isfinite(t).all() for t in (tr.xyz, tr.features, tr.opacity_raw, tr.scaling_raw, tr.rotation_raw))


def _two_rank_trainer_worker(rank, port, exchange, views, ret):
    """Two gloo ranks sharing the one GPU: each trains on its own view(s); the gradient exchange (factored or
    dense, one or two views per rank and step) must leave both parameter replicas bit-identical, and every rank
    must hold the same totals of the densification statistics without a separate merge of the sums."""
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=2)
    try:
        P, W, H, D = 2500, 96, 64, 3
        sc = synthetic.make_scene(P, D, (-3.8, -3.0), (1.3, 1.3, 1.3), seed=8, sh_coeffs=16)
        eyes = ((0.7, 0.5, -4.0), (-1.8, 0.4, -3.5), (1.6, -0.3, -3.7), (-0.4, 0.9, -3.9))
        cams = [make_camera(W, H, synthetic.FOVX, eyes[rank * views + k], device=DEV) for k in range(views)]
        gts = [torch.rand(3, H, W, generator=torch.Generator().manual_seed(60 + rank * views + k)).to(DEV)
               for k in range(views)]
        bg = torch.zeros(3, device=DEV)
        tr = train_step.GaussianTrainer(sc, DEV, factored_exchange=exchange == "factored", views_per_rank=views)
        assert tr.exchange.stats and not tr.fused
        cam = cams[0]
        for _ in range(3):
            tr.step(cams if views > 1 else cam, gts if views > 1 else gts[0], bg, math.tan(cam.FoVx * 0.5),
                    math.tan(cam.FoVy * 0.5))
        torch.cuda.synchronize()
        # the sums are global already; a Gaussian is counted once per view of either rank that rendered it
        seen = float(tr.denom.max())
        assert 0 < seen <= 3 * 2 * views
        tr.sync_densification_stats()  # merges the largest radius only
        assert float(tr.denom.max()) == seen
        torch.cuda.synchronize()
        state = [t.detach().cpu() for t in (tr.xyz, tr.features, tr.opacity_raw, tr.scaling_raw, tr.rotation_raw,
                                            tr.xyz_gradient_accum, tr.denom, tr.max_radii2D.float())]
        flat = torch.cat([t.reshape(-1).double() for t in state])
        both = [torch.empty_like(flat) for _ in range(2)]
        dist.all_gather(both, flat)
        moved = float((tr.features.cpu() - sc.shs).abs().max()) > 0
        ret[rank] = bool(torch.equal(both[0], both[1]) and torch.isfinite(flat).all() and moved)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("exchange,views", [("factored", 1), ("dense", 1), ("factored", 2), ("dense", 2)])
def test_two_rank_training_keeps_replicas_identical(exchange, views, C):
    import os
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    ret = mgr.dict()
    port = free_port()
    mp.spawn(_two_rank_trainer_worker, args=(port, exchange, views, ret), nprocs=2, join=True)
    assert dict(ret) == {0: True, 1: True}


def test_view_batches_on_one_rank_match_both_exchange_forms(C):
    """Two views per step on ONE rank (no process group): the factored and the dense bookkeeping must take the same
    training trajectory (up to the rounding of float atomics), and the statistics must count both views."""
    P, W, H, D = 3000, 128, 80, 3
    sc = synthetic.make_scene(P, D, (-3.8, -3.0), (1.3, 1.3, 1.3), seed=9, sh_coeffs=16)
    cams = [make_camera(W, H, synthetic.FOVX, e, device=DEV) for e in ((0.7, 0.5, -4.0), (-1.8, 0.4, -3.5))]
    gts = [torch.rand(3, H, W, generator=torch.Generator().manual_seed(80 + k)).to(DEV) for k in range(2)]
    bg = torch.zeros(3, device=DEV)
    tanx, tany = math.tan(cams[0].FoVx * 0.5), math.tan(cams[0].FoVy * 0.5)
    trs = [train_step.GaussianTrainer(sc, DEV, factored_exchange=f, views_per_rank=2) for f in (True, False)]
    for tr in trs:
        assert tr.batched and not tr.fused
        for _ in range(3):
            tr.step(cams, gts, bg, tanx, tany)
    a, b = trs
    lr_of = dict(xyz=a.opt.position_lr_init, features=a.opt.feature_lr, opacity_raw=a.opt.opacity_lr,
                 scaling_raw=a.opt.scaling_lr, rotation_raw=a.opt.rotation_lr)
    for n, lr in lr_of.items():
        diff = (getattr(a, n) - getattr(b, n)).abs()
        assert float((diff <= 1e-3 * lr).float().mean()) > 0.995, n
        assert float(diff.max()) <= 6.5 * lr, n
    assert torch.equal(a.denom, b.denom) and float(a.denom.max()) == 6.0
    assert torch.equal(a.max_radii2D, b.max_radii2D) and int(a.max_radii2D.max()) > 0
    assert torch.allclose(a.xyz_gradient_accum, b.xyz_gradient_accum, rtol=1e-3, atol=1e-9)
    with pytest.raises(RuntimeError):
        a.step(cams[0], gts[0], bg, tanx, tany)


# ---- densification, pruning, opacity reset -----------------------------------------------------------------------

class _TorchAdamScene:
    """The same state held the reference's way -- one nn.Parameter per group inside torch.optim.Adam -- with the
    reference's surgery on it (scene/gaussian_model.py:394-548): rows appended with zero moments, rows dropped with
    their moments, a replaced tensor whose moments restart.  Built from a trainer so both start from equal bits."""
    GROUPS = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")

    def __init__(self, tr, percent_dense):
        cl = lambda t: torch.nn.Parameter(t.detach().clone())
        vals = dict(xyz=tr.xyz, f_dc=tr.features[:, :1], f_rest=tr.features[:, 1:], opacity=tr.opacity_raw,
                    scaling=tr.scaling_raw, rotation=tr.rotation_raw)
        mom = dict(xyz=tr.state["xyz"], opacity=tr.state["opacity_raw"], scaling=tr.state["scaling_raw"],
                   rotation=tr.state["rotation_raw"],
                   f_dc=tuple(t[:, :1] for t in tr.state["features"]), f_rest=tuple(t[:, 1:] for t in tr.state["features"]))
        self.opt = torch.optim.Adam([{"params": [cl(vals[n])], "lr": 0.0, "name": n} for n in self.GROUPS], eps=1e-15)
        for g in self.opt.param_groups:
            m, v = mom[g["name"]]
            self.opt.state[g["params"][0]] = dict(step=torch.tensor(5.0), exp_avg=m.clone().contiguous(),
                                                  exp_avg_sq=v.clone().contiguous())
        self.accum, self.denom = tr.xyz_gradient_accum.clone(), tr.denom.clone()
        self.max_radii = tr.max_radii2D.clone()
        self.percent_dense = percent_dense

    def p(self, name):
        return next(g["params"][0] for g in self.opt.param_groups if g["name"] == name)

    def scaling(self):
        return torch.exp(self.p("scaling"))

    def _swap(self, transform_param, transform_moment):
        for g in self.opt.param_groups:
            old = g["params"][0]
            st = self.opt.state.pop(old)
            st["exp_avg"], st["exp_avg_sq"] = transform_moment(g["name"], st["exp_avg"]), transform_moment(g["name"], st["exp_avg_sq"])
            g["params"][0] = torch.nn.Parameter(transform_param(g["name"], old.detach()))
            self.opt.state[g["params"][0]] = st

    def append(self, new):
        self._swap(lambda n, t: torch.cat((t, new[n]), dim=0), lambda n, t: torch.cat((t, torch.zeros_like(new[n])), dim=0))
        n = self.p("xyz").shape[0]
        self.accum, self.denom = torch.zeros((n, 1), device=DEV), torch.zeros((n, 1), device=DEV)
        self.max_radii = torch.zeros(n, dtype=torch.int32, device=DEV)

    def prune(self, mask):
        keep = ~mask
        self._swap(lambda n, t: t[keep], lambda n, t: t[keep])
        self.accum, self.denom, self.max_radii = self.accum[keep], self.denom[keep], self.max_radii[keep]

    def densify_and_prune(self, max_grad, min_opacity, extent, max_screen_size, generator):
        from neuralgaussiansplatting_amd.gaussian_model import quaternion_to_rotation
        grads = self.accum / self.denom
        grads[grads.isnan()] = 0.0
        small = self.scaling().max(dim=1).values <= self.percent_dense * extent
        sel = (torch.norm(grads, dim=-1) >= max_grad) & small
        self.append({n: self.p(n).detach()[sel] for n in self.GROUPS})
        n0 = self.p("xyz").shape[0]
        padded = torch.zeros(n0, device=DEV)
        padded[:grads.shape[0]] = grads.squeeze()
        sel = (padded >= max_grad) & (self.scaling().max(dim=1).values > self.percent_dense * extent)
        stds = self.scaling().detach()[sel].repeat(2, 1)
        samples = torch.normal(mean=torch.zeros_like(stds), std=stds, generator=generator)
        rots = quaternion_to_rotation(self.p("rotation").detach()[sel]).repeat(2, 1, 1)
        new = {n: self.p(n).detach()[sel].repeat(2, *([1] * (self.p(n).dim() - 1))) for n in self.GROUPS}
        new["xyz"] = torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + new["xyz"]
        new["scaling"] = torch.log(stds / (0.8 * 2))
        self.append(new)
        self.prune(torch.cat((sel, torch.zeros(2 * int(sel.sum()), dtype=torch.bool, device=DEV))))
        mask = (torch.sigmoid(self.p("opacity").detach()) < min_opacity).squeeze()
        if max_screen_size:
            mask = mask | (self.max_radii > max_screen_size) | (self.scaling().detach().max(dim=1).values > 0.1 * extent)
        self.prune(mask)


def _small_trainer(P=3000, seed=21, log_scale=(-4.2, -2.6)):
    sc = synthetic.make_scene(P, 3, log_scale, (1.3, 1.3, 1.3), seed=seed, sh_coeffs=16)
    return train_step.GaussianTrainer(sc, DEV)


@pytest.mark.parametrize("max_screen_size", [None, 20])
def test_densify_and_prune_matches_optimizer_surgery(C, max_screen_size):
    """From one given state (parameters, Adam moments, statistics) the trainer's clone / split / prune must leave
    exactly what the reference's operations on torch.optim.Adam leave: same rows in the same order, same moments."""
    tr = _small_trainer()
    P = tr.P
    gen = torch.Generator().manual_seed(77)
    for name in tr.state:
        m, v = tr.state[name]
        m.copy_(torch.randn(m.shape, generator=gen) * 1e-3)
        v.copy_(torch.rand(v.shape, generator=gen) * 1e-5)
    tr.denom.copy_(torch.randint(0, 4, (P, 1), generator=gen).float())  # zeros give 0/0 -> NaN -> 0
    tr.xyz_gradient_accum.copy_(torch.rand(P, 1, generator=gen) * 1e-3 * tr.denom.cpu())
    tr.opacity_raw[::17] = -6.0  # sigmoid < 0.005: pruned
    tr._activate()
    extent = 2.0
    ref = _TorchAdamScene(tr, tr.opt.percent_dense)
    tr.max_radii2D.copy_(torch.randint(0, 40, (P,), generator=gen).int())  # cleared by the clone step on both sides
    ref.max_radii = tr.max_radii2D.clone()
    tr.extras["normals"] = torch.arange(P, device=DEV).float().reshape(P, 1).repeat(1, 3)
    g1 = torch.Generator(device=DEV).manual_seed(5)
    g2 = torch.Generator(device=DEV).manual_seed(5)
    tr.densify_and_prune(0.0002, 0.005, extent, max_screen_size, generator=g1)
    ref.densify_and_prune(0.0002, 0.005, extent, max_screen_size, generator=g2)
    assert tr.P == ref.p("xyz").shape[0] and tr.P != P
    want = dict(xyz=ref.p("xyz"), features=torch.cat((ref.p("f_dc"), ref.p("f_rest")), dim=1), opacity_raw=ref.p("opacity"),
                scaling_raw=ref.p("scaling"), rotation_raw=ref.p("rotation"))
    for name, w in want.items():
        assert torch.equal(getattr(tr, name), w.detach()), name
    st = lambda n, k: ref.opt.state[ref.p(n)][k]
    for k, i in (("exp_avg", 0), ("exp_avg_sq", 1)):
        assert torch.equal(tr.state["xyz"][i], st("xyz", k))
        assert torch.equal(tr.state["features"][i], torch.cat((st("f_dc", k), st("f_rest", k)), dim=1))
        assert torch.equal(tr.state["opacity_raw"][i], st("opacity", k))
        assert torch.equal(tr.state["scaling_raw"][i], st("scaling", k))
        assert torch.equal(tr.state["rotation_raw"][i], st("rotation", k))
    assert torch.equal(tr.xyz_gradient_accum, ref.accum) and torch.equal(tr.denom, ref.denom)
    assert torch.equal(tr.max_radii2D, ref.max_radii)
    # activated copies, gradient buffers and extra rows follow the new size
    assert torch.equal(tr.scales, torch.exp(tr.scaling_raw)) and tr.opacities.shape == (tr.P, 1)
    assert tr.exchange.P == tr.P and tr.extras["normals"].shape == (tr.P, 3)
    # a split Gaussian's two children carry their parent's extra row; clones too
    assert int(tr.extras["normals"][:, 0].unique().numel()) < tr.P
    # and the resized trainer keeps training
    cam = make_camera(96, 64, synthetic.FOVX, (0.7, 0.5, -4.0), device=DEV)
    gt = torch.rand(3, 64, 96, generator=gen).to(DEV)
    tr.step(cam, gt, torch.zeros(3, device=DEV), math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5))
    torch.cuda.synchronize()
    assert all(torch.isfinite(t).all() for t in (tr.xyz, tr.features, tr.opacity_raw, tr.scaling_raw, tr.rotation_raw))


def test_reset_opacity_caps_and_restarts_moments(C):
    tr = _small_trainer(500)
    tr.state["opacity_raw"][0].fill_(0.3)
    before = tr.opacities.clone()
    tr.reset_opacity()
    assert float(tr.opacities.max()) <= 0.01 + 1e-7
    low = before < 0.01
    assert torch.allclose(tr.opacities[low], before[low], atol=1e-7)
    assert float(tr.state["opacity_raw"][0].abs().max()) == 0.0 and float(tr.state["opacity_raw"][1].abs().max()) == 0.0
    assert torch.allclose(torch.sigmoid(tr.opacity_raw), tr.opacities)


def test_schedule_skips_the_steps_torch_adam_skips(C):
    """train_iteration on a compressed schedule, beside a torch.optim.Adam whose parameters are replaced at the
    points where the reference replaces them (train.py:113-128: densification replaces all, the opacity reset the
    opacities; a replaced parameter has no gradient when the optimiser runs, so its step count stands still)."""
    tr = _small_trainer(2500, seed=4, log_scale=(-3.9, -2.9))
    o = train_step.OptimizationParams()
    o.iterations, o.densify_from_iter, o.densification_interval = 14, 2, 3
    o.opacity_reset_interval, o.densify_until_iter = 6, 11
    tr.opt = o
    names = ("xyz", "features", "opacity_raw", "scaling_raw", "rotation_raw")
    dummy = {n: torch.nn.Parameter(torch.zeros(1)) for n in names}
    opt = torch.optim.Adam([{"params": [dummy[n]], "name": n} for n in names], lr=0.0)

    def replace(group_names):
        for g in opt.param_groups:
            if g["name"] in group_names:
                st = opt.state.pop(g["params"][0], None)
                g["params"][0] = torch.nn.Parameter(torch.zeros(1))
                if st is not None:
                    opt.state[g["params"][0]] = st

    cam = make_camera(96, 64, synthetic.FOVX, (0.7, 0.5, -4.0), device=DEV)
    gt = torch.rand(3, 64, 96, generator=torch.Generator().manual_seed(2)).to(DEV)
    bg = torch.ones(3, device=DEV)
    sizes, degree0 = [tr.P], tr.sh_degree
    for it in range(1, o.iterations + 1):
        for g in opt.param_groups:
            g["params"][0].grad = torch.ones(1)
        if it < o.densify_until_iter:
            if it > o.densify_from_iter and it % o.densification_interval == 0:
                replace(names)
            if it % o.opacity_reset_interval == 0 or it == o.densify_from_iter:  # white background
                replace(("opacity_raw",))
        if it < o.iterations:
            opt.step()
        opt.zero_grad(set_to_none=True)
        tr.train_iteration(cam, gt, bg, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), scene_extent=2.0,
                           white_background=True)
        sizes.append(tr.P)
        want = {g["name"]: int(opt.state[g["params"][0]]["step"]) if opt.state.get(g["params"][0]) else 0
                for g in opt.param_groups}
        assert tr.adam_steps == want, (it, tr.adam_steps, want)
    torch.cuda.synchronize()
    assert tr.iteration == o.iterations and tr.sh_degree == degree0
    assert len(set(sizes)) > 1  # the set of Gaussians did change
    assert tr.adam_steps["opacity_raw"] == tr.adam_steps["xyz"] - 1
    assert all(torch.isfinite(t).all() for t in (tr.xyz, tr.features, tr.opacity_raw, tr.scaling_raw, tr.rotation_raw))
    tr.sh_degree = 1
    tr.oneupSHdegree(); tr.oneupSHdegree(); tr.oneupSHdegree()
    assert tr.sh_degree == 3 == tr.max_sh_degree


def test_activated_adam_takes_a_step_count_per_group(C):
    gen = torch.Generator().manual_seed(9)
    P = 300
    raw0 = [torch.randn(P, w, generator=gen) for w in (1, 3, 4)]
    grads = [torch.randn(P, w, generator=gen) for w in (1, 3, 4)]
    mom = [(torch.randn(P, w, generator=gen) * 0.01, torch.rand(P, w, generator=gen) * 1e-4) for w in (1, 3, 4)]
    lrs, steps = (0.05, 0.005, 0.001), (3, 40, 41)

    def run(step_arg, only=None):
        raw = [r.clone().to(DEV) for r in raw0]
        m, v = [a.clone().to(DEV) for a, _ in mom], [b.clone().to(DEV) for _, b in mom]
        act = [torch.empty_like(r) for r in raw]
        C.adam_step_activated(raw, [g.to(DEV) for g in grads], m, v, act, lrs, step_arg)
        return raw
    mixed = run(steps)
    for i, t in enumerate(steps):
        assert torch.equal(mixed[i], run(t)[i])
    assert not torch.equal(mixed[0], run(steps[1])[0])
    with pytest.raises(RuntimeError):
        run((1, 0, 1))


def _two_rank_densify_worker(rank, port, ret):
    """Two gloo ranks sharing the one GPU run the full loop body on a compressed schedule, each on its own view:
    the merged statistics and the identically seeded sampler must lead both to the same new set of Gaussians."""
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=2)
    try:
        P, W, H, D = 2501, 96, 64, 3  # an odd count: gradient buffers must cope with any size
        sc = synthetic.make_scene(P, D, (-3.9, -2.9), (1.3, 1.3, 1.3), seed=4, sh_coeffs=16)
        cam = make_camera(W, H, synthetic.FOVX, ((0.7, 0.5, -4.0), (-1.8, 0.4, -3.5))[rank], device=DEV)
        gt = torch.rand(3, H, W, generator=torch.Generator().manual_seed(60 + rank)).to(DEV)
        bg = torch.zeros(3, device=DEV)
        tr = train_step.GaussianTrainer(sc, DEV)
        o = train_step.OptimizationParams()
        o.iterations, o.densify_from_iter, o.densification_interval = 9, 1, 3
        o.opacity_reset_interval, o.densify_until_iter = 6, 8
        tr.opt = o
        sizes = [tr.P]
        for _ in range(o.iterations):
            tr.train_iteration(cam, gt, bg, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), scene_extent=2.0, seed=3)
            sizes.append(tr.P)
        torch.cuda.synchronize()
        state = [t.detach().cpu() for t in (tr.xyz, tr.features, tr.opacity_raw, tr.scaling_raw, tr.rotation_raw)]
        flat = torch.cat([t.reshape(-1).double() for t in state] + [torch.tensor(sizes).double()])
        n = torch.tensor([flat.numel()])
        counts = [torch.zeros_like(n) for _ in range(2)]
        dist.all_gather(counts, n)
        same_size = int(counts[0]) == int(counts[1])
        equal = False
        if same_size:
            both = [torch.empty_like(flat) for _ in range(2)]
            dist.all_gather(both, flat)
            equal = torch.equal(both[0], both[1])
        ret[rank] = bool(same_size and equal and torch.isfinite(flat).all() and len(set(sizes)) > 1)
    finally:
        dist.destroy_process_group()


def test_two_rank_densification_keeps_replicas_identical(C):
    import os
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    ret = mgr.dict()
    port = free_port()
    mp.spawn(_two_rank_densify_worker, args=(port, ret), nprocs=2, join=True)
    assert dict(ret) == {0: True, 1: True}


def test_point_cloud_to_trained_ply(C, tmp_path):
    """The two ends around the trainer: a point cloud initialises a model (nearest-neighbour scales), the trainer
    takes it over, trains with densification on a compressed schedule, and the result is saved and read back."""
    from neuralgaussiansplatting_amd.gaussian_model import GaussianModel
    from neuralgaussiansplatting_amd.ply_io import BasicPointCloud
    gen = np.random.default_rng(3)
    n = 1500
    pts = (gen.random((n, 3)) * 2.0 - 1.0).astype(np.float32)
    pcd = BasicPointCloud(points=pts, colors=gen.random((n, 3)).astype(np.float32), normals=np.zeros((n, 3), np.float32))
    model = GaussianModel(3).create_from_pcd(pcd, spatial_lr_scale=2.0, device=DEV)
    tr = train_step.GaussianTrainer.from_model(model, DEV)
    assert tr.M == 16 and tr.sh_degree == 0 and tr.max_sh_degree == 3 and tr.spatial_lr_scale == 2.0
    assert torch.equal(tr.scaling_raw, model._scaling.detach()) and torch.equal(tr.opacity_raw, model._opacity.detach())
    o = train_step.OptimizationParams()
    o.iterations, o.densify_from_iter, o.densification_interval = 12, 2, 4
    o.opacity_reset_interval, o.densify_until_iter = 8, 11
    o.densify_grad_threshold = 1e-6  # a dozen iterations gather little gradient
    tr.opt = o
    cam = make_camera(96, 64, synthetic.FOVX, (0.7, 0.5, -4.0), device=DEV)
    gt = torch.rand(3, 64, 96, generator=torch.Generator().manual_seed(2)).to(DEV)
    for _ in range(o.iterations):
        tr.train_iteration(cam, gt, torch.zeros(3, device=DEV), math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5),
                           scene_extent=2.0)
    assert tr.P != n and tr.extras["features64"].shape == (tr.P, 64)
    out = tr.to_model()
    path = str(tmp_path / "point_cloud" / "iteration_12" / "point_cloud.ply")
    out.save_ply(path)
    back = GaussianModel(3).load_ply(path, device=DEV)
    for a, b in zip(out.parameters(), back.parameters()):
        assert torch.equal(a.detach(), b.detach())
    assert back._xyz.shape == (tr.P, 3) and torch.isfinite(back._xyz).all()


def test_capture_restore_resumes_bit_for_bit(C, tmp_path):
    """A checkpoint written mid-run (after densifications, so sizes and step counts have moved) and restored into
    a fresh trainer reproduces the whole state bit for bit; both then follow the same schedule.  (Their parameters
    drift apart at rounding level afterwards: float atomics add in a different order from run to run.)"""
    o = train_step.OptimizationParams()
    o.iterations, o.densify_from_iter, o.densification_interval = 20, 1, 3
    o.opacity_reset_interval, o.densify_until_iter = 6, 15
    cam = make_camera(96, 64, synthetic.FOVX, (0.7, 0.5, -4.0), device=DEV)
    gt = torch.rand(3, 64, 96, generator=torch.Generator().manual_seed(2)).to(DEV)
    bg = torch.zeros(3, device=DEV)
    run = lambda t: t.train_iteration(cam, gt, bg, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), scene_extent=2.0)
    a = _small_trainer(2000, seed=4, log_scale=(-3.9, -2.9))
    a.opt = o
    a.extras["normals"] = torch.zeros(a.P, 3, device=DEV)
    for _ in range(7):
        run(a)
    path = str(tmp_path / "chkpnt7.pth")
    torch.save((a.capture(), a.iteration), path)
    captured, it = torch.load(path, weights_only=True)
    b = _small_trainer(10, seed=1)
    b.opt = o
    b.restore(captured)
    assert it == 7 == b.iteration and b.P == a.P != 2000 and b.adam_steps == a.adam_steps
    for n in a._ROWS:
        assert torch.equal(getattr(a, n), getattr(b, n))
        assert torch.equal(a.state[n][0], b.state[n][0]) and torch.equal(a.state[n][1], b.state[n][1])
    assert torch.equal(a.xyz_gradient_accum, b.xyz_gradient_accum) and torch.equal(a.max_radii2D, b.max_radii2D)
    assert torch.equal(a.scales, b.scales) and b.extras["normals"].shape == (b.P, 3)
    for _ in range(5):
        run(a)
        run(b)
    torch.cuda.synchronize()
    assert a.iteration == b.iteration == 12 and a.adam_steps == b.adam_steps
    assert abs(a.P - b.P) <= max(4, a.P // 100)
    assert all(torch.isfinite(t).all() for t in (b.xyz, b.features, b.opacity_raw, b.scaling_raw, b.rotation_raw))


# NGS_RANDOM_CONFIGS=N widens the sweep for a soak run
@pytest.mark.parametrize("seed", range(int(os.environ.get("NGS_RANDOM_CONFIGS", "4"))))
def test_trainer_random_sizes(C, seed):
    """Any number of Gaussians (gradient buffers, Adam kernels and the fused SH step see unaligned sizes after
    every densification) and any image size: a compressed schedule must run through with finite parameters."""
    rng = np.random.RandomState(11000 + seed)
    P = int(rng.choice([rng.randint(1, 8), rng.randint(8, 300), rng.randint(300, 5000)]))
    W, H = int(rng.randint(8, 200)), int(rng.randint(8, 150))
    lo = float(rng.uniform(-4.5, -2.0))
    sc = synthetic.make_scene(P, 3, (lo, lo + 1.0), (1.3, 1.3, 1.3), seed=seed, sh_coeffs=16)
    tr = train_step.GaussianTrainer(sc, DEV)
    o = train_step.OptimizationParams()
    o.iterations, o.densify_from_iter, o.densification_interval = 8, 1, 2
    o.opacity_reset_interval, o.densify_until_iter = 4, 8
    o.densify_grad_threshold = float(10.0 ** rng.uniform(-7, -3))
    tr.opt = o
    tr.sh_degree = int(rng.randint(0, 4))
    cam = make_camera(W, H, synthetic.FOVX, (float(rng.uniform(-1, 1)), 0.4, float(rng.uniform(-4.5, -2.5))), device=DEV)
    gt = torch.rand(3, H, W, generator=torch.Generator().manual_seed(seed)).to(DEV)
    bg = torch.rand(3, generator=torch.Generator().manual_seed(seed)).to(DEV)
    for _ in range(o.iterations):
        terms, color, radii = tr.train_iteration(cam, gt, bg, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5),
                                                 scene_extent=float(rng.uniform(0.5, 5.0)),
                                                 white_background=bool(seed % 2))
        assert color.shape == (3, H, W)
    torch.cuda.synchronize()
    assert torch.isfinite(terms).all()
    for t in (tr.xyz, tr.features, tr.opacity_raw, tr.scaling_raw, tr.rotation_raw):
        assert t.shape[0] == tr.P and torch.isfinite(t).all()
    assert tr.opacities.shape == (tr.P, 1) and tr.exchange.P == tr.P


def test_activated_adam_aligned_and_unaligned_paths_agree(C):
    """16-byte aligned arrays take the float4 kernel, anything else the scalar one: same bits either way."""
    gen = torch.Generator().manual_seed(21)
    for P in (1, 2, 3, 5, 64, 259, 1002):
        data = [[torch.randn(P * w + 1, generator=gen) for w in (1, 3, 4)] for _ in range(4)]  # raw, grad, m, v
        data[3] = [t.abs() * 1e-3 for t in data[3]]
        results = []
        for shift in (0, 1):  # shift 1: opacity and scale arrays start 4 bytes off a 16-byte boundary
            groups = []
            for arrays in data:
                views = []
                for t, w in zip(arrays, (1, 3, 4)):
                    if w == 4 or shift == 0:
                        views.append(t[:P * w].clone().to(DEV).view(P, w))
                    else:
                        views.append(torch.cat([torch.zeros(1), t[:P * w]]).to(DEV)[1:].view(P, w))
                groups.append(views)
            act = [torch.empty(P * w + 4, device=DEV)[shift if w != 4 else 0:][:P * w].view(P, w) for w in (1, 3, 4)]
            assert all(v.is_contiguous() for grp in groups for v in grp)
            C.adam_step_activated(groups[0], groups[1], groups[2], groups[3], act, (0.05, 0.005, 0.001), (3, 4, 4))
            results.append([t.clone() for t in groups[0] + groups[2] + groups[3] + act])
        for a, b in zip(*results):
            assert torch.equal(a, b), P


def _raw_scene(config, seed=5):
    if config in ("c2", "c3"):
        sc, W, H = synthetic.make_config(config)
        shs = sc.shs
        if shs.shape[1] != 16:  # the raw entry points take any coefficient count; the fused step wants full rows
            shs = torch.cat([shs, torch.zeros(shs.shape[0], 16 - shs.shape[1], 3)], dim=1)
        sc = synthetic.Scene(sc.means3D, shs.contiguous(), sc.opacities, sc.scales, sc.rotations, sc.sh_degree)
    else:
        P, W, H = 5000, 200, 120
        sc = synthetic.make_scene(P, 3, (-4.0, -2.8), (1.3, 1.3, 1.3), seed=seed, sh_coeffs=16)
    g = torch.Generator().manual_seed(seed)
    raw = dict(opacity=train_step.inverse_sigmoid(sc.opacities), scaling=torch.log(sc.scales),
               rotation=sc.rotations * (0.5 + torch.rand(sc.rotations.shape[0], 1, generator=g) * 2.0))  # not unit length
    return sc.to(DEV), {k: v.to(DEV).contiguous() for k, v in raw.items()}, W, H


@pytest.mark.parametrize("config", ["small", "c2", "c3"])
def test_raw_parameter_path_equals_activated_path(C, config):
    """rasterize_gaussians_raw on (inverse-sigmoid opacity, log scale, unnormalised quaternion) against
    rasterize_gaussians on the activated values: image, radii and instance count bit-equal; the raw gradients equal
    the activated ones pushed through the activations' backward in float32, operation by operation."""
    sc, raw, W, H = _raw_scene(config)
    P, D = sc.means3D.shape[0], sc.sh_degree
    cam = make_camera(W, H, synthetic.FOVX, (0.7, 0.5, -4.0), device=DEV)
    tanx, tany = math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5)
    bg, e = torch.tensor([0.1, 0.2, 0.3], device=DEV), torch.empty(0)
    op, scl, rot = C.activate_parameters(raw["opacity"], raw["scaling"], raw["rotation"])
    assert torch.allclose(op, torch.sigmoid(raw["opacity"]), rtol=1e-6, atol=1e-7)
    assert torch.allclose(scl, torch.exp(raw["scaling"]), rtol=1e-6)
    assert torch.allclose(rot, torch.nn.functional.normalize(raw["rotation"]), rtol=1e-6, atol=1e-7)
    view = (cam.world_view_transform, cam.full_proj_transform, tanx, tany)
    a = C.rasterize_gaussians(bg, sc.means3D, e, op, scl, rot, 1.0, e, *view, H, W, sc.shs, D, cam.camera_center,
                              False, False)
    r = C.rasterize_gaussians_raw(bg, sc.means3D, e, raw["opacity"], raw["scaling"], raw["rotation"], 1.0, *view, H, W,
                                  sc.shs, D, cam.camera_center, False, False)
    assert a[0] == r[0] and a[0] > 0
    assert torch.equal(a[1], r[1]) and torch.equal(a[2], r[2])
    dL = torch.randn(3, H, W, generator=torch.Generator().manual_seed(9)).to(DEV)
    ga = C.rasterize_gaussians_backward(bg, sc.means3D, a[2], e, scl, rot, 1.0, e, *view, dL, sc.shs, D,
                                        cam.camera_center, a[3], a[0], a[4], a[5], False)
    gr = C.rasterize_gaussians_backward_raw(bg, sc.means3D, r[2], e, raw["opacity"], raw["scaling"], raw["rotation"],
                                            1.0, *view, dL, sc.shs, D, cam.camera_center, r[3], r[0], r[4], r[5], False)
    names = ("dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales",
             "dL_drotations")
    ga, gr = dict(zip(names, ga)), dict(zip(names, gr))
    # float atomics land in a different order in the two compositing passes: gradients agree to rounding, not bits
    def close(x, y, what):
        big = float(y.abs().max())
        assert float((x - y).abs().max()) <= 2e-5 * big + 1e-12, what
    for k in ("dL_dmeans2D", "dL_dcolors", "dL_dmeans3D", "dL_dcov3D", "dL_dsh"):
        close(gr[k], ga[k], k)
    close(gr["dL_dopacity"], ga["dL_dopacity"] * ((1.0 - op) * op), "opacity")
    close(gr["dL_dscales"], ga["dL_dscales"] * scl, "scales")
    g = ga["dL_drotations"]
    t = rot * g
    dot = ((t[:, 0] + t[:, 1]) + t[:, 2]) + t[:, 3]
    s = raw["rotation"] * raw["rotation"]
    length = torch.sqrt(((s[:, 0] + s[:, 1]) + s[:, 2]) + s[:, 3]).clamp_min(1e-12)
    close(gr["dL_drotations"], (g - rot * dot[:, None]) / length[:, None], "rotations")
    # against autograd through torch's own activations (what the reference differentiates)
    leaf = {k: v.clone().requires_grad_(True) for k, v in raw.items()}
    acts = (torch.sigmoid(leaf["opacity"]), torch.exp(leaf["scaling"]), torch.nn.functional.normalize(leaf["rotation"]))
    torch.autograd.backward(acts, (ga["dL_dopacity"], ga["dL_dscales"], ga["dL_drotations"]))
    for k, name in (("opacity", "dL_dopacity"), ("scaling", "dL_dscales"), ("rotation", "dL_drotations")):
        big = float(leaf[k].grad.abs().max())
        assert float((gr[name] - leaf[k].grad).abs().max()) <= 1e-4 * big, k
    # culled Gaussians get exact zeros
    culled = r[2] == 0
    assert int(culled.sum()) > 0 or config == "small"
    for k in names:
        assert float(gr[k][culled].abs().sum()) == 0.0, k


@pytest.mark.parametrize("config", ["small", "c3"])
def test_fused_training_equals_separate_optimiser_kernels(C, config):
    """The optimiser step inside the backward pass (raw parameters in and out, no gradient in memory) against the
    same iterations with gradients written out and separate Adam kernels reading activated copies: parameters,
    moments and densification statistics stay BIT-identical only if both see the same gradients, and the float
    atomics of two compositing passes need not agree in order -- so the comparison is at rounding level on the
    first step and, where Adam's sign-like first steps amplify a rounding-level gradient, within a few learning
    rates for a handful of elements afterwards."""
    if config == "c3":
        sc, W, H = synthetic.make_config("c3")
    else:
        W, H = 200, 120
        sc = synthetic.make_scene(6000, 3, (-4.0, -2.8), (1.3, 1.3, 1.3), seed=8, sh_coeffs=16)
    cams = [make_camera(W, H, synthetic.FOVX, e, device=DEV) for e in ((0.7, 0.5, -4.0), (-2.0, 0.4, -3.4), (1.5, -0.3, -3.6))]
    gts = [torch.rand(3, H, W, generator=torch.Generator().manual_seed(70 + i)).to(DEV) for i in range(3)]
    bg = torch.zeros(3, device=DEV)
    a = train_step.GaussianTrainer(sc, DEV, fused=True)
    b = train_step.GaussianTrainer(sc, DEV, fused=False)
    assert a.fused and not b.fused
    rows = ("xyz", "features", "opacity_raw", "scaling_raw", "rotation_raw")
    lr_of = dict(xyz=a.opt.position_lr_init, features=a.opt.feature_lr, opacity_raw=a.opt.opacity_lr,
                 scaling_raw=a.opt.scaling_lr, rotation_raw=a.opt.rotation_lr)
    for it in range(4):
        cam, gt = cams[it % 3], gts[it % 3]
        tanx, tany = math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5)
        hold = it == 2  # an opacity-reset iteration: the opacities keep their values, their moments still move
        ta, ca, ra = a.step(cam, gt, bg, tanx, tany, hold_opacity=hold)
        tb, cb, rb = b.step(cam, gt, bg, tanx, tany, hold_opacity=hold)
        if it == 0:
            assert torch.equal(ca, cb) and torch.equal(ra, rb) and torch.equal(ta, tb)  # same parameters, same image
        for n in rows:
            pa, pb = getattr(a, n), getattr(b, n)
            diff = (pa - pb).abs()
            assert float((diff <= 1e-3 * lr_of[n]).float().mean()) > 0.995, (it, n)
            assert float(diff.max()) <= 6.5 * lr_of[n], (it, n)
            for ma, mb in zip(a.state[n], b.state[n]):
                big = float(mb.abs().max())
                assert float((ma - mb).abs().max()) <= 1e-3 * big + 1e-20, (it, n)
        if hold:
            assert a.adam_steps == b.adam_steps and a.adam_steps["opacity_raw"] == it
    # the two trajectories drift apart by rounding-level gradient differences amplified by Adam's first steps (see
    # above), so a few Gaussians may cross a visibility or radius threshold on one side only
    assert float((a.denom == b.denom).float().mean()) > 0.999
    assert float((a.max_radii2D == b.max_radii2D).float().mean()) > 0.99
    rel = (a.xyz_gradient_accum - b.xyz_gradient_accum).abs() / (b.xyz_gradient_accum.abs() + 1e-7)
    assert float((rel < 2e-2).float().mean()) > 0.99
    # the activated copies follow on demand
    op, scl, rot = C.activate_parameters(a.opacity_raw, a.scaling_raw, a.rotation_raw)
    assert torch.equal(a.opacities, op) and torch.equal(a.scales, scl) and torch.equal(a.rotations, rot)


def test_fused_optimiser_step_is_bit_exact_given_the_same_gradients(C):
    """A scene whose compositing sums have ONE order only: every Gaussian is a speck whose visible pixels (alpha >=
    1/255) lie inside a single 4x4 pixel quadrant, so each of its gradient sums is reduced inside one 16-lane row
    and reaches memory as a single atomic.  Both ways of stepping then see bit-identical gradients, and because the
    optimiser arithmetic is shared code, parameters, moments and statistics must stay bit-identical throughout."""
    W, H, D = 256, 192, 3
    P = (W // 32) * (H // 32)
    g = torch.Generator().manual_seed(3)
    cam = make_camera(W, H, synthetic.FOVX, (0.0, 0.0, -4.0), device=DEV)
    tanx, tany = math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5)
    gx, gy = torch.meshgrid(torch.arange(W // 32), torch.arange(H // 32), indexing="ij")
    # pixel (32 i + 17, 32 j + 17): second pixel of a quadrant whichever way the camera's axes point
    px, py = 32.0 * gx.reshape(-1) + 17.0, 32.0 * gy.reshape(-1) + 17.0
    means = torch.stack([((2 * px + 1) / W - 1) * tanx * 4.0, ((2 * py + 1) / H - 1) * tany * 4.0, torch.zeros(P)], dim=1)
    sc = synthetic.Scene(means, torch.randn(P, 16, 3, generator=g) * 0.3, torch.full((P, 1), 0.5),
                         0.002 * (1 + 0.2 * torch.rand(P, 3, generator=g)),
                         torch.nn.functional.normalize(torch.randn(P, 4, generator=g)), D)
    a = train_step.GaussianTrainer(sc, DEV, fused=True)
    b = train_step.GaussianTrainer(sc, DEV, fused=False)
    gt = torch.rand(3, H, W, generator=g).to(DEV)
    bg = torch.zeros(3, device=DEV)
    rows = ("xyz", "features", "opacity_raw", "scaling_raw", "rotation_raw")
    for it in range(4):
        ta, ca, ra = a.step(cam, gt, bg, tanx, tany, hold_opacity=it == 2)
        tb, cb, rb = b.step(cam, gt, bg, tanx, tany, hold_opacity=it == 2)
        assert int((ra > 0).sum()) == P and int(ra.max()) <= 3
        assert torch.equal(ca, cb) and torch.equal(ta, tb), it
        for n in rows:
            assert torch.equal(getattr(a, n), getattr(b, n)), (it, n)
            assert torch.equal(a.state[n][0], b.state[n][0]) and torch.equal(a.state[n][1], b.state[n][1]), (it, n)
    assert not torch.equal(a.xyz, sc.means3D.to(DEV))
    assert torch.equal(a.xyz_gradient_accum, b.xyz_gradient_accum) and float(a.xyz_gradient_accum.sum()) > 0
    assert torch.equal(a.denom, b.denom) and torch.equal(a.max_radii2D, b.max_radii2D)


def test_view_statistics_rows(C):
    """ngs_view_statistics: the increments of xyz_gradient_accum / denom (scene/gaussian_model.py:550-552) as two
    summable rows, zeros where nothing was rendered, and the running maximum of the radii (train.py:115)."""
    g = torch.Generator().manual_seed(4)
    for P in (1, 63, 1000, 4097):
        g2d = torch.randn(P, 3, generator=g).to(DEV)
        radii = torch.randint(-1, 30, (P,), generator=g, dtype=torch.int32).to(DEV)
        big = torch.randint(0, 25, (P,), generator=g, dtype=torch.int32).to(DEV)
        norm, vis = torch.full((P,), 7.0, device=DEV), torch.full((P,), 7.0, device=DEV)
        want_big = torch.where(radii > 0, torch.maximum(big, radii), big)
        C.view_statistics(g2d, radii, norm, vis, big)
        seen = radii > 0
        assert torch.equal(vis, seen.float())
        want = torch.where(seen, torch.sqrt(g2d[:, 0] * g2d[:, 0] + g2d[:, 1] * g2d[:, 1]), torch.zeros((), device=DEV))
        assert torch.allclose(norm, want, rtol=1e-6, atol=0)
        assert float(norm[~seen].abs().sum()) == 0.0
        assert torch.equal(big, want_big)
        C.view_statistics(g2d, radii, norm, vis)  # without the maximum
    with pytest.raises(RuntimeError):
        C.view_statistics(g2d, radii.long(), norm, vis)


def test_raw_and_fused_entry_points_reject_bad_arguments(C):
    """Error paths of the raw-parameter entry points through the C ABI: every one returns an error (-> RuntimeError)
    instead of launching."""
    import ctypes
    sc, raw, W, H = _raw_scene("small")
    P = sc.means3D.shape[0]
    cam = make_camera(W, H, synthetic.FOVX, (0.7, 0.5, -4.0), device=DEV)
    tanx, tany = math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5)
    bg, e = torch.zeros(3, device=DEV), torch.empty(0)
    view = (cam.world_view_transform, cam.full_proj_transform, tanx, tany)
    with pytest.raises(RuntimeError, match="scales"):
        C.rasterize_gaussians_raw(bg, sc.means3D, e, raw["opacity"], e, raw["rotation"], 1.0, *view, H, W, sc.shs, 3,
                                  cam.camera_center, False, False)
    r = C.rasterize_gaussians_raw(bg, sc.means3D, e, raw["opacity"], raw["scaling"], raw["rotation"], 1.0, *view, H, W,
                                  sc.shs, 3, cam.camera_center, False, False)
    dL = torch.randn(3, H, W, generator=torch.Generator().manual_seed(1)).to(DEV)
    with pytest.raises(RuntimeError, match="raw"):
        C.rasterize_gaussians_backward_raw(bg, sc.means3D, r[2], e, e, raw["scaling"], raw["rotation"], 1.0, *view, dL,
                                           sc.shs, 3, cam.camera_center, r[3], r[0], r[4], r[5], False)
    z = lambda t: torch.zeros_like(t)
    params = dict(xyz=sc.means3D.clone(), features=sc.shs.clone(), opacity=raw["opacity"].clone(),
                  scaling=raw["scaling"].clone(), rotation=raw["rotation"].clone())
    groups = {k: (p, z(p), z(p), 1e-3, 1) for k, p in params.items()}
    call = lambda grp, **kw: C.rasterize_gaussians_backward_train(
        bg, r[2], 1.0, *view, dL, 3, cam.camera_center, r[3], r[0], r[4], r[5], grp, 1e-4, **kw)
    bad = dict(groups)
    bad["xyz"] = groups["xyz"][:4] + (0,)  # step 0
    with pytest.raises(RuntimeError, match="step"):
        call(bad)
    bad = dict(groups)
    bad["scaling"] = (params["scaling"], z(params["scaling"])[:-1], z(params["scaling"]), 1e-3, 1)
    with pytest.raises(RuntimeError, match="scaling"):
        call(bad)
    with pytest.raises(RuntimeError, match="Adam"):
        call(groups, betas=(1.0, 0.999))
    L = C.lib()
    st = C._TrainState()
    assert L.ngs_rasterize_backward_train(P, 3, r[0], None, W, H, ctypes.c_float(1.0), None, None, None,
                                          ctypes.c_float(tanx), ctypes.c_float(tany), None, None, None, None, None,
                                          ctypes.byref(st), 0, None) != 0
    assert b"missing" in L.ngs_last_error()
    assert L.ngs_rasterize_backward_train(P, 3, r[0], None, W, H, ctypes.c_float(1.0), None, None, None,
                                          ctypes.c_float(tanx), ctypes.c_float(tany), None, None, None, None, None, None,
                                          0, None) != 0
    # nothing moved
    assert torch.equal(params["xyz"], sc.means3D) and torch.equal(params["opacity"], raw["opacity"])
    # and a good call does move everything
    call(groups, densify=(torch.zeros(P, 1, device=DEV), torch.zeros(P, 1, device=DEV),
                          torch.zeros(P, dtype=torch.int32, device=DEV)))
    torch.cuda.synchronize()
    assert not torch.equal(params["xyz"], sc.means3D) and not torch.equal(params["features"], sc.shs)
