"""A three-Gaussian scene whose image and gradients are computed from first principles in float64, without the CPU
checker and without any restatement of the reference's kernels: a pinhole camera, EWA splatting
(covariance' = J R_cam Sigma R_cam^T J^T + 0.3 I), front-to-back alpha blending, and central finite differences of
that closed form for the gradients.  Anything that shares a misreading between the checker and the HIP kernels
(a transposed Jacobian, a swapped conic entry, a wrong pixel-centre convention, a wrong blending order) moves these
numbers by far more than the tolerances the tests use.

What the closed form takes from the reference are its documented constants only: the +0.3 low-pass on the
screen-space covariance, the 1/255 alpha threshold, the pixel-centre convention pix = ((ndc + 1) * S - 1) / 2 and
the degree-0 SH constant.  The scene is chosen so that the reference's other conventions stay inactive: opacities
at most 0.3 (alpha never reaches the 0.99 cap, and beyond three standard deviations alpha is below 1/255 anyway, so
the extent of the tile rectangle does not matter), three splats (transmittance never reaches the 1e-4 stop),
everything well inside the frustum guard band and in front of the near plane, unit quaternions.
"""
import math

import numpy as np
import torch

from neuralgaussiansplatting_amd import synthetic
from neuralgaussiansplatting_amd.cameras import make_camera

W, H = 48, 32
EYE = (0.4, -0.3, -4.0)
SH_C0 = 0.28209479177387814
BG = np.array([0.15, 0.4, 0.05])

PARAMS = dict(
    means=np.array([[0.10, 0.05, 0.0], [-0.25, 0.12, 0.6], [0.3, -0.2, -0.5]]),
    scales=np.array([[0.12, 0.05, 0.08], [0.2, 0.15, 0.04], [0.06, 0.18, 0.1]]),
    quats=np.array([[0.9, 0.1, -0.3, 0.2], [0.5, -0.5, 0.4, 0.3], [0.2, 0.7, 0.1, -0.6]]),  # (w, x, y, z)
    opacities=np.array([0.3, 0.22, 0.27]),
    sh_dc=np.array([[0.9, -0.4, 0.3], [-0.8, 1.1, 0.2], [0.1, 0.5, -1.2]]),
)
PARAMS["quats"] = PARAMS["quats"] / np.linalg.norm(PARAMS["quats"], axis=1, keepdims=True)


def camera():
    return make_camera(W, H, synthetic.FOVX, EYE)


def _rotation(q):
    """Rotation matrix of a quaternion (w, x, y, z), the textbook formula, used as given (no normalisation): the
    reference differentiates with respect to the quaternion it is handed."""
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def closed_form(p, cam=None, return_keep=False):
    """float64 image [3, H, W] of the scene ``p`` (dict like PARAMS)."""
    cam = cam or camera()
    V = cam.world_view_transform.double().numpy().T  # math form: p_cam = V @ [p, 1]
    Rc, tc = V[:3, :3], V[:3, 3]
    tanx, tany = math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2)
    fx, fy = W / (2 * tanx), H / (2 * tany)
    n = p["means"].shape[0]
    pc = p["means"] @ Rc.T + tc
    order = np.argsort(pc[:, 2], kind="stable")
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float64)
    img = np.zeros((3, H, W))
    T = np.ones((H, W))
    keeps = []
    for i in order:
        x, y, z = pc[i]
        cx, cy = fx * x / z + (W - 1) / 2, fy * y / z + (H - 1) / 2
        J = np.array([[fx / z, 0, -fx * x / z ** 2], [0, fy / z, -fy * y / z ** 2]])
        R = _rotation(p["quats"][i])
        Sigma = R @ np.diag(p["scales"][i] ** 2) @ R.T
        cov = J @ Rc @ Sigma @ Rc.T @ J.T + 0.3 * np.eye(2)
        inv = np.linalg.inv(cov)
        dx, dy = cx - xs, cy - ys
        power = -0.5 * (inv[0, 0] * dx * dx + 2 * inv[0, 1] * dx * dy + inv[1, 1] * dy * dy)
        alpha = p["opacities"][i] * np.exp(power)
        keep = alpha >= 1.0 / 255.0
        keeps.append(keep)
        alpha = np.where(keep, alpha, 0.0)
        color = np.maximum(SH_C0 * p["sh_dc"][i] + 0.5, 0.0)
        img += color[:, None, None] * (alpha * T)[None]
        T = T * (1 - alpha)
    img += BG[:, None, None] * T[None]
    return (img, keeps) if return_keep else img


def loss_weights():
    return np.random.RandomState(12).randn(3, H, W)


def finite_difference_gradients(h=1e-6):
    """{name: d(sum(weights * image))/d(param)} by central differences of the closed form.  The alpha threshold
    makes the image piecewise smooth; the perturbed evaluations must keep every pixel on its side of it."""
    wts = loss_weights()
    base, keep0 = closed_form(PARAMS, return_keep=True)
    out = {}
    for name in ("means", "scales", "quats", "opacities", "sh_dc"):
        g = np.zeros_like(PARAMS[name])
        for idx in np.ndindex(*PARAMS[name].shape):
            vals = []
            for sgn in (+1, -1):
                q = {k: v.copy() for k, v in PARAMS.items()}
                q[name][idx] += sgn * h
                img, keep = closed_form(q, return_keep=True)
                assert all((a == b).all() for a, b in zip(keep, keep0)), "a pixel crossed the alpha threshold"
                vals.append(float((img * wts).sum()))
            g[idx] = (vals[0] - vals[1]) / (2 * h)
        out[name] = g
    return base, out


def as_case():
    """The scene in the form tests/helpers.py cases have (float32 CPU tensors)."""
    f = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32).contiguous()
    shs = f(PARAMS["sh_dc"]).reshape(3, 1, 3)
    sc = synthetic.Scene(f(PARAMS["means"]), shs, f(PARAMS["opacities"]).reshape(3, 1), f(PARAMS["scales"]),
                         f(PARAMS["quats"]), 0)
    cam = camera()
    return dict(scene=sc, cam=cam, W=W, H=H, D=0, bg=f(BG), tanfovx=math.tan(cam.FoVx * 0.5),
                tanfovy=math.tan(cam.FoVy * 0.5))


GRAD_NAMES = {"means": "dL_dmeans3D", "scales": "dL_dscales", "quats": "dL_drotations", "opacities": "dL_dopacity",
              "sh_dc": "dL_dsh"}


def check_against_closed_form(image, grads, img_atol=2e-5, grad_rel=2e-4):
    """``image`` [3,H,W] and ``grads`` {rasterizer gradient name: array} of an implementation under test."""
    want_img, want = finite_difference_gradients()
    assert float(want_img.max()) > 0.3 and float(np.abs(want_img - BG[:, None, None]).max()) > 0.1
    np.testing.assert_allclose(image, want_img, rtol=0, atol=img_atol)
    for name, ref in want.items():
        got = np.asarray(grads[GRAD_NAMES[name]], dtype=np.float64).reshape(ref.shape)
        big = float(np.abs(ref).max())
        assert big > 0, name
        assert float(np.abs(got - ref).max()) <= grad_rel * big, (name, got, ref)
