"""An independent, differentiable PyTorch (float64, CPU) formulation of the
rasterizer's forward pass.  Gradients come from autograd, not from hand-written
formulas, so agreement with the CPU checker's backward pass validates the
gradient math itself, not just a restatement of it.

The discrete decisions (which Gaussians a tile lists and in what order, which
(pixel, Gaussian) pairs are skipped, where a pixel stops) are taken from the
checker's forward state and treated as constants, exactly as the reference's
backward pass treats them (backward.cu:486-501).  Two reference conventions are
reproduced on purpose: the 0.99 alpha cap is not differentiated
(backward.cu:499,538: the gradient flows as if unclamped), and quaternions are
used un-normalised (forward.cu:127).  A third convention concerns Gaussians
outside the 1.3x frustum guard band: the forward pass replaces the view-space
x (y) that enters the projection Jacobian by the clamped value
(forward.cu:80-85), and the backward pass treats that value as a constant --
no gradient reaches the position through it, neither via x nor via z
(backward.cu:175-176,262-264).  ``clamp_like_reference`` states exactly that
with a detach; inside the band it is the identity.
"""
import torch

C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
      1.445305721320277, -0.5900435899266435)


def sh_color(deg, shs, dirs):
    x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
    res = C0 * shs[:, 0]
    if deg > 0:
        res = res - C1 * y * shs[:, 1] + C1 * z * shs[:, 2] - C1 * x * shs[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        res = (res + C2[0] * xy * shs[:, 4] + C2[1] * yz * shs[:, 5] + C2[2] * (2 * zz - xx - yy) * shs[:, 6] +
               C2[3] * xz * shs[:, 7] + C2[4] * (xx - yy) * shs[:, 8])
    if deg > 2:
        res = (res + C3[0] * y * (3 * xx - yy) * shs[:, 9] + C3[1] * xy * z * shs[:, 10] +
               C3[2] * y * (4 * zz - xx - yy) * shs[:, 11] + C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * shs[:, 12] +
               C3[4] * x * (4 * zz - xx - yy) * shs[:, 13] + C3[5] * z * (xx - yy) * shs[:, 14] +
               C3[6] * x * (xx - 3 * yy) * shs[:, 15])
    return torch.clamp_min(res + 0.5, 0.0)


def rotation_unnormalised(q):
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    rows = [1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
            2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
            2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)]
    return torch.stack(rows, dim=1).view(-1, 3, 3)


def clamp_like_reference(t, tz, lim):
    """View-space x (or y) as the projection Jacobian sees it: unchanged inside the guard band, the band's edge
    times z -- as a constant -- outside it."""
    ratio = t / tz
    outside = (ratio < -lim) | (ratio > lim)
    edge = (torch.clamp(ratio, -lim, lim) * tz).detach()
    return torch.where(outside, edge, t)


def render(means3D, ndc_offset, shs, opacities, scales, rotations, view_t, full_t, campos, bg, W, H, tanx, tany, deg,
           state, scale_modifier=1.0):
    """-> image [3,H,W] (float64).  ``state`` is the checker's forward dict (lists, ranges, n_contrib).
    ``ndc_offset`` [P,2] is a zero tensor added to the projected position in NDC units; its gradient is what the
    rasterizer reports as the gradient of the screen-space means."""
    dt = torch.float64
    P = means3D.shape[0]
    view_t, full_t, campos, bg = view_t.to(dt), full_t.to(dt), campos.to(dt), bg.to(dt)
    hom = torch.cat([means3D, torch.ones(P, 1, dtype=dt)], dim=1)
    p_view = hom @ view_t
    p_hom = hom @ full_t
    p_w = 1.0 / (p_hom[:, 3:4] + 1e-7)
    ndc = p_hom[:, :2] * p_w + ndc_offset
    xy = torch.stack([((ndc[:, 0] + 1.0) * W - 1.0) * 0.5, ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5], dim=1)

    Rm = rotation_unnormalised(rotations)
    L = Rm * (scale_modifier * scales).unsqueeze(1)
    Sigma = L @ L.transpose(1, 2)
    fx, fy = W / (2.0 * tanx), H / (2.0 * tany)
    tz = p_view[:, 2]
    tx = clamp_like_reference(p_view[:, 0], tz, 1.3 * tanx)
    ty = clamp_like_reference(p_view[:, 1], tz, 1.3 * tany)
    zero = torch.zeros_like(tz)
    J = torch.stack([torch.stack([fx / tz, zero, -(fx * tx) / (tz * tz)], dim=1),
                     torch.stack([zero, fy / tz, -(fy * ty) / (tz * tz)], dim=1)], dim=1)  # [P,2,3]
    Wc = view_t[:3, :3].transpose(0, 1)  # world -> camera rotation
    A = J @ Wc
    cov2 = A @ Sigma @ A.transpose(1, 2)
    a = cov2[:, 0, 0] + 0.3
    b = cov2[:, 0, 1]
    c = cov2[:, 1, 1] + 0.3
    det = a * c - b * b
    conA, conB, conC = c / det, -b / det, a / det

    dirs = means3D - campos
    dirs = dirs / dirs.norm(dim=1, keepdim=True)
    rgb = sh_color(deg, shs, dirs)

    img = torch.zeros(3, H, W, dtype=dt)
    gx = (W + 15) // 16
    plist = torch.from_numpy(state["point_list"].astype("int64"))
    n_contrib = torch.from_numpy(state["n_contrib"].astype("int64"))
    for tile in range(state["ranges"].shape[0]):
        lo, hi = int(state["ranges"][tile, 0]), int(state["ranges"][tile, 1])
        ty0, tx0 = (tile // gx) * 16, (tile % gx) * 16
        ys = torch.arange(ty0, min(ty0 + 16, H))
        xs = torch.arange(tx0, min(tx0 + 16, W))
        if len(ys) == 0 or len(xs) == 0:
            continue
        py, px = torch.meshgrid(ys, xs, indexing="ij")
        py, px = py.reshape(-1), px.reshape(-1)
        if hi == lo:
            img[:, py, px] = bg.view(3, 1).expand(3, len(py))
            continue
        ids = plist[lo:hi]
        dx = xy[ids, 0].unsqueeze(0) - px.to(dt).unsqueeze(1)  # [pix, K]
        dy = xy[ids, 1].unsqueeze(0) - py.to(dt).unsqueeze(1)
        power = -0.5 * (conA[ids] * dx * dx + conC[ids] * dy * dy) - conB[ids] * dx * dy
        G = torch.exp(power)
        raw = opacities[ids, 0] * G
        alpha = raw + (torch.clamp_max(raw, 0.99) - raw).detach()  # cap applied, not differentiated
        k = torch.arange(1, hi - lo + 1).unsqueeze(0)  # 1-based position in the tile list
        # a pair contributes iff it passes both skips and lies at or before the pixel's last contributor
        keep = (power.detach() <= 0) & (alpha.detach() >= 1.0 / 255.0) & (k <= n_contrib[py, px].unsqueeze(1))
        alpha_eff = torch.where(keep, alpha, torch.zeros_like(alpha))
        one_minus = 1.0 - alpha_eff
        T_before = torch.cumprod(torch.cat([torch.ones(len(py), 1, dtype=dt), one_minus[:, :-1]], dim=1), dim=1)
        wgt = alpha_eff * T_before
        T_final = T_before[:, -1] * one_minus[:, -1]
        col = wgt @ rgb[ids]  # [pix, 3]
        img[:, py, px] = (col + T_final.unsqueeze(1) * bg.view(1, 3)).transpose(0, 1)
    return img


def grads(scene, cam, state, dL_dpix, W, H, tanx, tany, deg, bg, scale_modifier=1.0):
    """Autograd gradients (float64) named like the rasterizer's outputs."""
    dt = torch.float64
    leaf = lambda t: t.detach().to(dt).clone().requires_grad_(True)
    means, shs, opac, scales, rots = map(leaf, (scene.means3D, scene.shs, scene.opacities, scene.scales,
                                                scene.rotations))
    off = torch.zeros(means.shape[0], 2, dtype=dt, requires_grad=True)
    img = render(means, off, shs, opac, scales, rots, cam.world_view_transform, cam.full_proj_transform,
                 cam.camera_center, bg, W, H, tanx, tany, deg, state, scale_modifier)
    (img * dL_dpix.to(dt)).sum().backward()
    return img.detach(), dict(dL_dmeans3D=means.grad, dL_dmeans2D=off.grad, dL_dsh=shs.grad, dL_dopacity=opac.grad,
                              dL_dscales=scales.grad, dL_drotations=rots.grad)


def in_guard_band(scene, cam, tanx, tany):
    """True if no Gaussian is outside the 1.3x frustum clamp (or behind the near plane)."""
    hom = torch.cat([scene.means3D, torch.ones(scene.means3D.shape[0], 1)], dim=1) @ cam.world_view_transform
    z = hom[:, 2]
    ok = (z > 0.2) & ((hom[:, 0] / z).abs() < 1.29 * tanx) & ((hom[:, 1] / z).abs() < 1.29 * tany)
    return bool(ok.all())
