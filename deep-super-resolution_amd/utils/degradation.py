"""The reference's degradations (utils/degradation.py:5-20) on device-resident uint8 images, computed by HIP kernels.

Same function names and argument meaning as the reference; an image here is a ``torch.uint8`` tensor ``[H, W, 3]`` on the
MI355X (what ``np.array(PIL image)`` holds on the host).  Numpy arrays and PIL images are accepted too -- they are uploaded,
processed on the device and returned in the type they came in -- so the reference's call sites keep working.

Bit-exactness: ``downsample`` reproduces Pillow's 8-bit bicubic resampler (fixed-point tables built here exactly like
libImaging/Resample.c's precompute_coeffs, the two passes run in csrc/data.hip); the noise functions draw from numpy's global
generator in the reference's order by default (``rng="numpy"``: same pixels as the reference for the same seed) or on the
device (``rng="device"``: torch's generator, for throughput).
"""
import ctypes as C
import math

import numpy as np
import torch

from .. import _lib
from ..functional import _need_gpu, _ptr, _stream, check

PRECISION_BITS = 32 - 8 - 2
_tables = {}


def _bicubic(x):
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def resample_tables(in_size, out_size, device):
    """(ksize, bounds int32 [out][2], kk int32 [out][ksize]) of Pillow's bicubic resampler for one axis, on `device` (cached).
    Host float64 arithmetic in Pillow's order (precompute_coeffs, normalize_coeffs_8bpc); uploaded once per (in, out)."""
    key = (in_size, out_size, str(device))
    hit = _tables.get(key)
    if hit is not None:
        return hit
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    inv = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [_bicubic((x + xmin - center + 0.5) * inv) for x in range(xmax)]
        total = 0.0
        for v in w:
            total += v
        for x, v in enumerate(w):
            if total != 0.0:
                v = v / total
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    out = (ksize, torch.from_numpy(bounds).to(device), torch.from_numpy(kk).to(device))
    _tables[key] = out
    return out


def _to_device(image, device=None):
    """-> (uint8 [H,W,C] device tensor, restore(tensor) -> the caller's type)."""
    if torch.is_tensor(image):
        _need_gpu(image)
        if image.dtype != torch.uint8 or image.dim() != 3:
            raise TypeError(f"image tensor must be uint8 [H, W, C], got {image.dtype} {tuple(image.shape)}")
        return image.contiguous(), (lambda t: t)
    dev = torch.device(device if device is not None else "cuda:0")
    if isinstance(image, np.ndarray):
        if image.dtype != np.uint8 or image.ndim != 3:
            raise TypeError(f"image array must be uint8 [H, W, C], got {image.dtype} {image.shape}")
        return torch.from_numpy(np.ascontiguousarray(image)).to(dev), (lambda t: t.cpu().numpy())
    from PIL import Image                                            # a PIL image (what the reference's downsample takes)
    arr = np.array(image.convert("RGB"))
    return torch.from_numpy(arr).to(dev), (lambda t: Image.fromarray(t.cpu().numpy()))


def resize(image, out_w, out_h):
    """``PIL.Image.resize((out_w, out_h), Image.BICUBIC)`` (dataset.py:38-45) on the device."""
    img, restore = _to_device(image)
    h, w, c = img.shape
    lib = _lib.lib()
    out = img
    if out_w != w:
        ksize, bounds, kk = resample_tables(w, out_w, img.device)
        dst = torch.empty((h, out_w, c), dtype=torch.uint8, device=img.device)
        check(lib.dsr_resample_u8(_ptr(out), _ptr(dst), h, w, c, 1, out_w, _ptr(bounds), _ptr(kk), ksize, _stream()))
        out = dst
    if out_h != h:
        ksize, bounds, kk = resample_tables(h, out_h, img.device)
        dst = torch.empty((out_h, out.shape[1], c), dtype=torch.uint8, device=img.device)
        check(lib.dsr_resample_u8(_ptr(out), _ptr(dst), h, out.shape[1], c, 0, out_h, _ptr(bounds), _ptr(kk), ksize, _stream()))
        out = dst
    return restore(out)


def downsample(image, factor=2, interpolation=None):
    """utils/degradation.py:19-20: bicubic resize to (W // factor, H // factor).  `interpolation` other than bicubic is refused
    (the reference only ever passes its default)."""
    if interpolation is not None:
        from PIL import Image
        if interpolation != Image.BICUBIC:
            raise NotImplementedError("only Image.BICUBIC (the reference's default) is implemented on the device")
    if torch.is_tensor(image) or isinstance(image, np.ndarray):
        h, w = image.shape[0], image.shape[1]
    else:
        w, h = image.width, image.height
    return resize(image, w // factor, h // factor)


def add_gaussian_noise(image, std=1, rng="numpy"):
    """utils/degradation.py:5-7: clip(image + N(0, (std*255)^2), 0, 255) truncated to uint8.  rng="numpy": the normal draw comes
    from numpy's global generator (float64, same call as the reference: same result for the same seed); "device": torch."""
    img, restore = _to_device(image)
    if rng == "numpy":
        noise = torch.from_numpy(np.random.normal(scale=std * 255, size=tuple(img.shape))).to(img.device)   # float64
    elif rng == "device":
        noise = torch.randn(tuple(img.shape), dtype=torch.float32, device=img.device) * float(std * 255)
    else:
        raise ValueError("rng must be 'numpy' or 'device'")
    out = torch.empty_like(img)
    check(_lib.lib().dsr_noise_gaussian_u8(_ptr(img), _ptr(noise), int(noise.dtype == torch.float64), _ptr(out), img.numel(),
                                           _stream()))
    return restore(out)


def add_salt_pepper_noise(image, s=0.01, p=0.01, rng="numpy"):
    """utils/degradation.py:9-17: salt (255) where rand < s, then pepper (0) where rand < p, per pixel over all channels.
    Returns a new image (the reference writes into its argument and returns it)."""
    img, restore = _to_device(image)
    h, w, c = img.shape
    if rng == "numpy":
        salt = torch.from_numpy(np.random.rand(h, w) < s).to(img.device)
        pepper = torch.from_numpy(np.random.rand(h, w) < p).to(img.device)
    elif rng == "device":
        salt = torch.rand((h, w), device=img.device) < s
        pepper = torch.rand((h, w), device=img.device) < p
    else:
        raise ValueError("rng must be 'numpy' or 'device'")
    salt, pepper = salt.to(torch.uint8).contiguous(), pepper.to(torch.uint8).contiguous()
    out = torch.empty_like(img)
    check(_lib.lib().dsr_salt_pepper_u8(_ptr(img), _ptr(salt), _ptr(pepper), _ptr(out), h, w, c, _stream()))
    return restore(out)
