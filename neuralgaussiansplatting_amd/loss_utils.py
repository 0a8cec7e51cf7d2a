"""Image losses of the training step the rasterizer serves: L1 and SSIM with an
11x11 Gaussian window (sigma 1.5) applied as a depthwise convolution, as in the
reference's utils/loss_utils.py:17-63; train.py:91-92 combines them as
``(1 - 0.2) * L1 + 0.2 * (1 - SSIM)``."""
import torch
import torch.nn.functional as F


def l1_loss(network_output, gt):
    return (network_output - gt).abs().mean()


def l2_loss(network_output, gt):
    return ((network_output - gt) ** 2).mean()


def _window(size, channels, sigma=1.5):
    x = torch.arange(size, dtype=torch.float32) - size // 2
    g = torch.exp(-(x ** 2) / (2.0 * sigma ** 2))
    g = (g / g.sum()).unsqueeze(1)
    w2d = (g @ g.t()).unsqueeze(0).unsqueeze(0)
    return w2d.expand(channels, 1, size, size).contiguous()


def ssim(img1, img2, window_size=11, size_average=True):
    ch = img1.size(-3)
    w = _window(window_size, ch).to(device=img1.device, dtype=img1.dtype)
    pad = window_size // 2
    conv = lambda t: F.conv2d(t, w, padding=pad, groups=ch)
    mu1, mu2 = conv(img1), conv(img2)
    mu1_sq, mu2_sq, mu12 = mu1 * mu1, mu2 * mu2, mu1 * mu2
    s1 = conv(img1 * img1) - mu1_sq
    s2 = conv(img2 * img2) - mu2_sq
    s12 = conv(img1 * img2) - mu12
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    m = ((2 * mu12 + c1) * (2 * s12 + c2)) / ((mu1_sq + mu2_sq + c1) * (s1 + s2 + c2))
    return m.mean() if size_average else m.mean(1).mean(1).mean(1)


def training_loss(image, gt_image, lambda_dssim=0.2):
    return (1.0 - lambda_dssim) * l1_loss(image, gt_image) + lambda_dssim * (1.0 - ssim(image, gt_image))
