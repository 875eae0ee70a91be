"""Oracle (test infrastructure): the reference's data-side arithmetic, restated with numpy.

  resize_u8            PIL ``Image.resize(size, Image.BICUBIC)`` on an RGB uint8 image -- what the reference's
                       ``downsample`` (utils/degradation.py:19-20) and ``get_image_pair`` (dataset.py:21-45) call.  Pillow is a
                       third-party dependency of the reference; its 8-bit resampling (libImaging/Resample.c: float64 filter
                       weights, normalised, rounded to 22-bit fixed point, integer accumulation with a rounding offset,
                       horizontal pass then vertical pass, each clipped to uint8) is restated here and pinned BIT FOR BIT against
                       Pillow itself (tests/test_oracle_golden.py, live where Pillow is importable, and tests/golden/data_*.npz
                       produced by the reference's own utils/degradation.py).
  add_gaussian_noise   utils/degradation.py:5-7   (float64 add, clip, truncating uint8 cast)
  add_salt_pepper      utils/degradation.py:9-17  (salt first, pepper wins where both hit)
  to_tensor            torchvision ``ToTensor`` on a uint8 HWC array: float32 CHW / 255 (dataset.py:59-60; torchvision is absent
                       here -- formula restated, "parity unpinned")
  scale_images         dataset.py:149-159: LR /= 255 (a SECOND time, on top of ToTensor's), HR = HR / 255 * 2 - 1
  train_patch_coords   dataset.py:121-147: the two ``np.random.randint`` draws (x first, then y) and the patch edges

All integer / byte work: the HIP path (csrc/data.hip) has to match these functions bit for bit.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def bicubic_filter(x):
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def resample_coeffs(in_size, out_size):
    """Pillow's precompute_coeffs + normalize_coeffs_8bpc for the bicubic filter over the whole axis.
    Returns (ksize, bounds int32 [out_size][2] = (first input index, tap count), kk int32 [out_size][ksize])."""
    support_f = 2.0
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = support_f * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)              # (C cast: truncation; the operand is > -1 here wherever it matters)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [bicubic_filter((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        if ww != 0.0:
            w = [v / ww for v in w]
        for x, v in enumerate(w):
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return ksize, bounds, kk


def _pass(img, bounds, kk, axis):
    """One resampling pass along `axis` (0 = rows / vertical, 1 = columns / horizontal) of an [H][W][C] uint8 array."""
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((bounds.shape[0],) + src.shape[1:], dtype=np.uint8)
    for i in range(bounds.shape[0]):
        x0, n = int(bounds[i, 0]), int(bounds[i, 1])
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), dtype=np.int64)
        for t in range(n):
            acc += src[x0 + t] * int(kk[i, t])
        out[i] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)       # (arithmetic shift, then the clip8 table)
    return np.moveaxis(out, 0, axis)


def resize_u8(img, out_w, out_h):
    """``Image.fromarray(img).resize((out_w, out_h), Image.BICUBIC)`` for an [H][W][3] uint8 array."""
    h, w = img.shape[:2]
    out = img
    if out_w != w:
        _, b, k = resample_coeffs(w, out_w)
        out = _pass(out, b, k, 1)
    if out_h != h:
        _, b, k = resample_coeffs(h, out_h)
        out = _pass(out, b, k, 0)
    return out


def downsample(img, factor=2):
    """utils/degradation.py:19-20 on an array: size (W // factor, H // factor)."""
    return resize_u8(img, img.shape[1] // factor, img.shape[0] // factor)


def add_gaussian_noise(img, noise):
    """utils/degradation.py:5-7 with the normal draw made by the caller: ``noise = np.random.normal(scale=std*255, size=img.shape)``."""
    return np.clip(img + noise, 0, 255).astype(np.uint8)


def add_salt_pepper(img, salt, pepper):
    """utils/degradation.py:9-17 with the two uniform draws thresholded by the caller (``rand(H, W) < s`` / ``< p``)."""
    out = img.copy()
    out[salt] = 255
    out[pepper] = 0
    return out


def to_tensor(img):
    """uint8 [H][W][C] -> float32 [C][H][W] in [0, 1] (torchvision.transforms.ToTensor, dataset.py:59-60)."""
    return np.ascontiguousarray(img.transpose(2, 0, 1)).astype(np.float32) / np.float32(255)


def scale_images(lr, hr):
    """dataset.py:149-159 on the float32 CHW arrays of to_tensor (in-place ops of the reference, as expressions)."""
    lr = lr / np.float32(255.0)
    hr = hr / np.float32(255.0)
    hr = hr * np.float32(2)
    hr = hr - np.float32(1)
    return lr, hr


def train_patch_coords(lr_h, lr_w, patch_w, patch_h, scale, rng):
    """dataset.py:121-147: (LR top, LR left, HR top, HR left) from the reference's two randint draws (x, then y).
    `rng` is a ``np.random.RandomState`` (the reference uses the global one)."""
    cx = rng.randint(patch_w // 2, lr_w - patch_w // 2)
    cy = rng.randint(patch_h // 2, lr_h - patch_h // 2)
    left = int(cx - patch_w // 2)
    top = int(cy - patch_h // 2)
    return top, left, top * scale, left * scale


def sample_image(name, h, w):
    """Deterministic RGB uint8 test image [h][w][3]: smooth structure + hash noise (regenerated wherever needed, never stored)."""
    from . import filler
    u = filler.tensor(name, (h, w, 3)).numpy().astype(np.float64)           # uniform(-1, 1)
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([np.sin(xx / 7.0) * np.cos(yy / 11.0), np.cos((xx + yy) / 13.0), np.sin(yy / 5.0)], -1)
    return np.clip(128 + 90 * base + 40 * u, 0, 255).astype(np.uint8)
