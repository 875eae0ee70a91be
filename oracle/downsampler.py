"""Oracle (test infrastructure): restatement of the fixed-kernel Downsampler.

Follows /root/reference/utils/downsampler.py:
  :9-63    Downsampler.__init__ (kernel selection, dense diagonal Conv2d, ReplicationPad2d)
  :65-71   forward
  :73-135  get_kernel (lanczos / gauss / box), float64, normalised to sum 1 (:133)
The reference realises the filter as Conv2d(n, n, k, stride=factor) whose only non-zero
filters are the n diagonal ones (:44-50) and whose bias is zero, i.e. a depthwise
strided correlation -- which is what is computed here.
"""
import numpy as np
import torch
import torch.nn.functional as F


def resolve(factor, kernel_type, kernel_width=None, support=None, sigma=None):
    """downsampler.py:14-38 -> (kernel_type_, kernel_width, support, sigma)."""
    if kernel_type == "lanczos2":
        return "lanczos", 4 * factor + 1, 2, sigma
    if kernel_type == "lanczos3":
        return "lanczos", 6 * factor + 1, 3, sigma
    if kernel_type == "gauss12":
        return "gauss", 7, support, 1 / 2
    if kernel_type == "gauss1sq2":
        return "gauss", 9, support, 1.0 / np.sqrt(2)
    if kernel_type in ("lanczos", "gauss", "box"):
        return kernel_type, kernel_width, support, sigma
    raise AssertionError("wrong name kernel")


def get_kernel(factor, kernel_type, phase, kernel_width, support=None, sigma=None):
    """downsampler.py:73-135, vectorised; returns float64 [k, k]."""
    assert kernel_type in ("lanczos", "gauss", "box")
    if phase == 0.5 and kernel_type != "box":
        size = kernel_width - 1
    else:
        size = kernel_width
    if kernel_type == "box":
        assert phase == 0.5, "Box filter is always half-phased"
        kernel = np.full([size, size], 1.0 / (kernel_width * kernel_width))
    elif kernel_type == "gauss":
        assert sigma, "sigma is not specified"
        assert phase != 0.5, "phase 1/2 for gauss not implemented"
        center = (kernel_width + 1.0) / 2.0
        idx = np.arange(1, size + 1, dtype=np.float64)
        d = (idx - center) / 2.0
        sigma_sq = sigma * sigma
        kernel = np.exp(-(d[:, None] ** 2 + d[None, :] ** 2) / (2 * sigma_sq)) / (2.0 * np.pi * sigma_sq)
    else:
        assert support, "support is not specified"
        center = (kernel_width + 1) / 2.0
        idx = np.arange(1, size + 1, dtype=np.float64)
        d = np.abs(idx + 0.5 - center) / factor if phase == 0.5 else np.abs(idx - center) / factor
        with np.errstate(divide="ignore", invalid="ignore"):
            v = support * np.sin(np.pi * d) * np.sin(np.pi * d / support) / (np.pi * np.pi * d * d)
        v = np.where(d != 0, v, 1.0)
        kernel = v[:, None] * v[None, :]
    return kernel / kernel.sum()


def padding_of(kernel_size, factor):
    """downsampler.py:54-59."""
    if kernel_size % 2 == 1:
        return int((kernel_size - 1) / 2.0)
    return int((kernel_size - factor) / 2.0)


def downsampler_forward(x, factor, kernel_type, phase=0, kernel_width=None, support=None, sigma=None,
                        preserve_size=False):
    kt, kw, sup, sig = resolve(factor, kernel_type, kernel_width, support, sigma)
    k = get_kernel(factor, kt, phase, kw, support=sup, sigma=sig)
    kf = torch.from_numpy(k).to(torch.float32).to(x.dtype)   # float64 -> float32 weight copy (downsampler.py:48-50)
    if preserve_size:
        p = padding_of(k.shape[0], factor)
        x = F.pad(x, (p, p, p, p), mode="replicate")
    c = x.shape[1]
    w = kf[None, None].expand(c, 1, *kf.shape).contiguous()
    return F.conv2d(x, w, None, stride=factor, groups=c)
