"""Generates tests/golden/*.npz from the reference's own importable Python.

Run in the build container only (it imports from /root/reference, which does
not exist on the GPU box):  python tests/golden/make_golden.py

The reference ships no tests or golden vectors for the rasterizer, and its
CUDA sources cannot be built here, so these fixtures pin what its Python can
pin: SH -> RGB (utils/sh_utils.py:57-112, the function behind the
``convert_SHs_python`` switch, gaussian_renderer/__init__.py:74-79) and its
gradients under autograd, the
camera matrices (utils/graphics_utils.py:38-71 combined as in
scene/cameras.py:54-57), the training loss terms (utils/loss_utils.py:17-63), the
combined loss of train.py:90-92 with its gradient wrt the image, and the
position learning-rate schedule (utils/general_utils.py:29-61).
Only inputs and outputs are stored.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/reference")
from utils.graphics_utils import getProjectionMatrix, getWorld2View2  # noqa: E402
from utils.loss_utils import l1_loss, ssim  # noqa: E402
from utils.sh_utils import eval_sh  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def sh_fixture():
    g = torch.Generator().manual_seed(2024)
    P = 97
    means = (torch.rand(P, 3, generator=g) * 2 - 1) * 1.5
    campos = torch.tensor([0.3, -0.2, -3.5])
    shs = torch.randn(P, 16, 3, generator=g) * 2.0  # [P, M, 3] as the rasterizer takes them; large enough to clamp
    out = {"means": means.numpy(), "campos": campos.numpy(), "shs": shs.numpy()}
    dirs = means - campos
    dirs = dirs / dirs.norm(dim=1, keepdim=True)
    for deg in range(4):
        rgb = eval_sh(deg, shs.transpose(1, 2), dirs)  # reference layout [P, 3, M]
        out["rgb_deg%d" % deg] = torch.clamp_min(rgb + 0.5, 0.0).numpy()
        out["raw_deg%d" % deg] = (rgb + 0.5).numpy()
    np.savez(os.path.join(HERE, "sh_to_rgb.npz"), **out)


def sh_backward_fixture():
    """Gradients of the colour stage as the reference's ``convert_SHs_python`` branch computes them
    (gaussian_renderer/__init__.py:74-79): eval_sh on the normalised view direction, +0.5, clamp at 0, all under
    autograd, for a fixed dL/dcolour.  dL/dshs and dL/dmeans (through the direction's normalisation) are what the
    rasterizer's SH backward must reproduce (cuda_rasterizer/backward.cu:20-139, auxiliary.h:107-117)."""
    g = torch.Generator().manual_seed(4242)
    P = 131
    means0 = (torch.rand(P, 3, generator=g) * 2 - 1) * 1.5
    campos = torch.tensor([0.25, -0.4, -3.0])
    shs0 = torch.randn(P, 16, 3, generator=g) * 1.5  # large enough that some channels clamp
    dL_dcolor = torch.randn(P, 3, generator=g)
    out = {"means": means0.numpy(), "campos": campos.numpy(), "shs": shs0.numpy(), "dL_dcolor": dL_dcolor.numpy()}
    for deg in range(4):
        means = means0.clone().requires_grad_(True)
        shs = shs0.clone().requires_grad_(True)
        shs_view = shs.transpose(1, 2).view(-1, 3, 16)
        dir_pp = means - campos.repeat(P, 1)
        dir_pp_normalized = dir_pp / dir_pp.norm(dim=1, keepdim=True)
        sh2rgb = eval_sh(deg, shs_view, dir_pp_normalized)
        colors = torch.clamp_min(sh2rgb + 0.5, 0.0)
        (colors * dL_dcolor).sum().backward()
        out["raw_deg%d" % deg] = (sh2rgb + 0.5).detach().numpy()
        out["dL_dshs_deg%d" % deg] = shs.grad.numpy()
        # degree 0 does not look at the direction: autograd leaves no gradient on the positions
        out["dL_dmeans_deg%d" % deg] = np.zeros((P, 3), np.float32) if means.grad is None else means.grad.numpy()
        # The same map in factored form, so that a test can apply it to ANY colour gradient (the one a GPU run
        # produces): d(colour before the clamp)/d(shs) is the basis value per coefficient, and
        # jac[p, j, c] = d(colour c before the clamp of Gaussian p)/d(means[p, j]).
        jac = np.zeros((P, 3, 3), np.float32)
        basis = None
        for ch in range(3):
            m2 = means0.clone().requires_grad_(True)
            s2 = shs0.clone().requires_grad_(True)
            d2 = m2 - campos.repeat(P, 1)
            raw = eval_sh(deg, s2.transpose(1, 2).view(-1, 3, 16), d2 / d2.norm(dim=1, keepdim=True))
            raw[:, ch].sum().backward()
            if m2.grad is not None:
                jac[:, :, ch] = m2.grad.numpy()
            if ch == 0:
                basis = s2.grad[:, :, 0].numpy().copy()
        out["jac_deg%d" % deg] = jac
        out["basis_deg%d" % deg] = basis
    np.savez(os.path.join(HERE, "sh_backward.npz"), **out)


def camera_fixture():
    out = {}
    rng = np.random.RandomState(7)
    for i in range(4):
        # a random rotation (QR) and translation, FoVs in the range the datasets use
        q, _ = np.linalg.qr(rng.randn(3, 3))
        if np.linalg.det(q) < 0:
            q[:, 0] = -q[:, 0]
        t = rng.randn(3) * 2.0
        fovx, fovy = 0.5 + 0.2 * i, 0.35 + 0.15 * i
        view_t = torch.tensor(getWorld2View2(q, t, np.array([0.0, 0.0, 0.0]), 1.0)).transpose(0, 1)
        proj_t = getProjectionMatrix(znear=0.01, zfar=100.0, fovX=fovx, fovY=fovy).transpose(0, 1)
        full = view_t.unsqueeze(0).bmm(proj_t.unsqueeze(0)).squeeze(0)
        out["R%d" % i], out["t%d" % i] = q, t
        out["fov%d" % i] = np.array([fovx, fovy])
        out["view_t%d" % i] = view_t.numpy()
        out["proj_t%d" % i] = proj_t.numpy()
        out["full%d" % i] = full.numpy()
        out["campos%d" % i] = view_t.inverse()[3, :3].numpy()
    np.savez(os.path.join(HERE, "camera_matrices.npz"), **out)


def loss_fixture():
    g = torch.Generator().manual_seed(5)
    a = torch.rand(3, 40, 56, generator=g)
    b = (a + 0.1 * torch.randn(3, 40, 56, generator=g)).clamp(0, 1)
    np.savez(os.path.join(HERE, "loss_terms.npz"), img_a=a.numpy(), img_b=b.numpy(),
             l1=np.float64(l1_loss(a, b).item()), ssim=np.float64(ssim(a, b).item()))


def train_loss_fixture():
    """Loss of train.py:90-92 and its gradient wrt the rendered image, through the reference's own functions."""
    g = torch.Generator().manual_seed(11)
    a = torch.rand(3, 45, 70, generator=g, requires_grad=True)  # not a multiple of the 16-pixel tiles
    b = (a.detach() + 0.15 * torch.randn(3, 45, 70, generator=g)).clamp(0, 1)
    b[:, 10:20, 30:50] = a.detach()[:, 10:20, 30:50]  # a region of exact agreement: sign(0) = 0 in the L1 term
    lam = 0.2
    ll1 = l1_loss(a, b)
    s = ssim(a, b)
    loss = (1.0 - lam) * ll1 + lam * (1.0 - s)
    loss.backward()
    np.savez(os.path.join(HERE, "train_loss.npz"), img=a.detach().numpy(), gt=b.numpy(), lambda_dssim=np.float64(lam),
             l1=np.float64(ll1.item()), ssim=np.float64(s.item()), loss=np.float64(loss.item()), grad=a.grad.numpy())


def lr_fixture():
    """Position learning-rate schedule (utils/general_utils.py:29-61) with the defaults of arguments/__init__.py:74-77."""
    from utils.general_utils import get_expon_lr_func
    steps = np.array([0, 1, 2, 10, 100, 999, 1000, 7000, 15000, 29999, 30000, 40000], dtype=np.int64)
    f = get_expon_lr_func(lr_init=0.00016, lr_final=0.0000016, lr_delay_mult=0.01, max_steps=30_000)
    f2 = get_expon_lr_func(lr_init=0.01, lr_final=0.0001, lr_delay_steps=500, lr_delay_mult=0.05, max_steps=5000)
    np.savez(os.path.join(HERE, "expon_lr.npz"), steps=steps, default=np.array([f(int(s)) for s in steps]),
             delayed=np.array([f2(int(s)) for s in steps]))


class _zeros_on_cpu:
    """utils/general_utils.py:65,83,102 allocate their results with ``torch.zeros(..., device="cuda")``; there is no
    GPU where the fixtures are made, so for the duration of a call that one argument is redirected to the CPU.
    The arithmetic of the reference's functions is untouched."""

    def __enter__(self):
        self._zeros = torch.zeros

        def zeros(*a, **k):
            if str(k.get("device", "cpu")).startswith("cuda"):
                k["device"] = "cpu"
            return self._zeros(*a, **k)

        torch.zeros = zeros

    def __exit__(self, *exc):
        torch.zeros = self._zeros


def cov3d_fixture():
    """3D covariance from scale and rotation as the reference's ``compute_cov3D_python`` branch computes it
    (gaussian_renderer/__init__.py:63-64 -> scene/gaussian_model.py:29-33 -> utils/general_utils.py:64-110), and its
    derivatives under autograd.  This is what cuda_rasterizer/forward.cu:118-152 and backward.cu:278-341 restate.

    Stored per scale modifier: cov [P,6]; for a fixed dL/dcov the gradients dL/dscale and dL/drot; and the same
    map in factored form -- jac_s[p, j, k] = d cov[p, k] / d scale[p, j], jac_r[p, j, k] = d cov[p, k] / d rot[p, j]
    -- so that a test can apply it to the dL/dcov a GPU run produces.

    The reference's Python normalises the quaternion inside build_rotation, the kernel uses it as given; with unit
    quaternions as input the covariances agree, and the Python's rotation gradient is the kernel's with its
    component along the quaternion removed (the Jacobian of the normalisation at a unit quaternion is I - q q^T).
    That component never reaches a parameter: render() feeds normalize(_rotation), whose backward removes it too."""
    from utils.general_utils import build_scaling_rotation, strip_symmetric

    def build_covariance_from_scaling_rotation(scaling, scaling_modifier, rotation):  # scene/gaussian_model.py:29-33
        L = build_scaling_rotation(scaling_modifier * scaling, rotation)
        actual_covariance = L @ L.transpose(1, 2)
        return strip_symmetric(actual_covariance)

    g = torch.Generator().manual_seed(777)
    P = 211
    scales0 = torch.exp(torch.rand(P, 3, generator=g) * 5.0 - 5.5)  # 0.004 .. 0.6, axis ratios up to ~150
    rots0 = torch.nn.functional.normalize(torch.randn(P, 4, generator=g))
    rots0[0] = torch.tensor([1.0, 0.0, 0.0, 0.0])
    rots0[1] = torch.tensor([0.0, 1.0, 0.0, 0.0])
    dL_dcov = torch.randn(P, 6, generator=g)
    out = {"scales": scales0.numpy(), "rotations": rots0.numpy(), "dL_dcov": dL_dcov.numpy(),
           "modifiers": np.array([1.0, 0.6, 1.7], np.float32)}
    with _zeros_on_cpu():
        for mi, mod in enumerate([1.0, 0.6, 1.7]):
            s = scales0.clone().requires_grad_(True)
            r = rots0.clone().requires_grad_(True)
            cov = build_covariance_from_scaling_rotation(s, mod, r)
            (cov * dL_dcov).sum().backward()
            out["cov_m%d" % mi] = cov.detach().numpy()
            out["dL_dscale_m%d" % mi] = s.grad.numpy()
            out["dL_drot_m%d" % mi] = r.grad.numpy()
            jac_s = np.zeros((P, 3, 6), np.float32)
            jac_r = np.zeros((P, 4, 6), np.float32)
            for k in range(6):
                s2 = scales0.clone().requires_grad_(True)
                r2 = rots0.clone().requires_grad_(True)
                build_covariance_from_scaling_rotation(s2, mod, r2)[:, k].sum().backward()
                jac_s[:, :, k] = s2.grad.numpy()
                jac_r[:, :, k] = r2.grad.numpy()
            out["jac_s_m%d" % mi] = jac_s
            out["jac_r_m%d" % mi] = jac_r
    np.savez(os.path.join(HERE, "cov3d.npz"), **out)


if __name__ == "__main__":
    if sys.argv[1:] == ["cov3d"]:
        cov3d_fixture()
        sys.exit(0)
    sh_fixture()
    sh_backward_fixture()
    camera_fixture()
    loss_fixture()
    train_loss_fixture()
    lr_fixture()
    cov3d_fixture()
    print("wrote", sorted(f for f in os.listdir(HERE) if f.endswith(".npz")))
