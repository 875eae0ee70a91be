"""Oracle (test infrastructure): image-quality metrics of the reference's logging / evaluation code.

Call sites in /root/reference: train_GAN.py:30-32,110-112; DIP.py:73-75,157-159,183-185; eval_GAN.py:30-32,47-49
(``PSNR()``, ``SSIM(data_range=1.)`` of torchmetrics).  torchmetrics is not installed here and cannot be fetched, and the
reference holds no fixture for these numbers: PARITY UNPINNED.  Both are restated from their published definitions:
  PSNR  10 log10(range^2 / MSE)                                  (see oracle/losses.py)
  SSIM  Wang, Bovik, Sheikh, Simoncelli 2004, as torchmetrics configures it by default: Gaussian 11x11 window, sigma 1.5,
        K1 = 0.01, K2 = 0.03, computed per channel; torchmetrics reflects-pads by 5 and crops that border again, i.e. the
        mean runs over the window positions that lie inside the image.
"""
import torch
import torch.nn.functional as F


def gaussian_window(size=11, sigma=1.5, dtype=torch.float64):
    d = torch.arange(size, dtype=dtype) - (size - 1) / 2.0
    g = torch.exp(-d * d / (2.0 * sigma * sigma))
    return g / g.sum()


def ssim(a, b, data_range=1.0, size=11, sigma=1.5, k1=0.01, k2=0.03):
    """Mean SSIM of two [N,C,H,W] tensors (float64 arithmetic)."""
    a, b = a.double(), b.double()
    n, c, h, w = a.shape
    g = gaussian_window(size, sigma)
    win = (g[:, None] * g[None, :])[None, None].expand(c, 1, size, size)
    mu_a, mu_b = F.conv2d(a, win, groups=c), F.conv2d(b, win, groups=c)
    s_aa = F.conv2d(a * a, win, groups=c) - mu_a * mu_a
    s_bb = F.conv2d(b * b, win, groups=c) - mu_b * mu_b
    s_ab = F.conv2d(a * b, win, groups=c) - mu_a * mu_b
    c1, c2 = (k1 * data_range) ** 2, (k2 * data_range) ** 2
    m = ((2 * mu_a * mu_b + c1) * (2 * s_ab + c2)) / ((mu_a * mu_a + mu_b * mu_b + c1) * (s_aa + s_bb + c2))
    return float(m.mean())
