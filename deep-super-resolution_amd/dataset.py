"""The reference's data path (dataset.py:9-159) with the per-image arithmetic on the MI355X.

  get_image_pair / DIV2KDataset / GANDIV2KDataset   same names, constructor arguments and return values as the reference; file
        decoding stays Pillow on the host, everything after it (the bicubic resizes, the noise, ToTensor, scale_images, the
        training patches) runs as HIP kernels on device-resident uint8 images.  Items come back as device tensors.
  PatchBank   what the reference's DataLoader + GANDIV2KDataset amount to for a training step, restructured for the device: the
        whole (pre-shrunk) image set lives in HBM as uint8 and ``sample(batch)`` cuts and converts a batch of LR / HR patch
        pairs in two kernel launches -- the step is fed at its own rate instead of the host's.

The reference scales by 255 twice (ToTensor at :59-60, then scale_images :152,155): ``reference_scaling=True`` (default)
reproduces that, bit for bit; ``False`` gives the [0,1] / [-1,1] ranges its comments describe (SURVEY.md 8f row 1 asks for
the choice to be explicit).  torchvision is not needed (ToTensor = /255 into CHW float32).
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from .functional import _ptr, _stream, check
from .utils import degradation

PATCH_UNIT, PATCH_LR_REF, PATCH_HR_REF, PATCH_HR_UNIT = range(4)


def _device(device):
    return torch.device(device if device is not None else "cuda:0")


def patch_batch(images, tops, lefts, ph, pw, mode):
    """fp32 [B,3,ph,pw] batch of patches, patch b cut from uint8 [H,W,3] device image images[b] at (tops[b], lefts[b])."""
    n = len(images)
    if not (n == len(tops) == len(lefts)) or n == 0:
        raise ValueError("patch_batch: images, tops and lefts must be equally long and non-empty")
    for im in images:
        if not (torch.is_tensor(im) and im.is_cuda and im.dtype == torch.uint8 and im.dim() == 3 and im.shape[2] == 3
                and im.is_contiguous()):
            raise TypeError("patch_batch: images must be contiguous uint8 [H, W, 3] tensors on the device")
    out = torch.empty((n, 3, ph, pw), dtype=torch.float32, device=images[0].device)
    ptrs = (C.c_void_p * n)(*[im.data_ptr() for im in images])
    ints = lambda v: (C.c_int * n)(*[int(q) for q in v])
    check(_lib.lib().dsr_patch_batch_u8(n, ptrs, ints([im.shape[0] for im in images]), ints([im.shape[1] for im in images]),
                                        ints(tops), ints(lefts), ph, pw, mode, _ptr(out), _stream()))
    return out


def to_tensor(image):
    """uint8 [H,W,3] device image -> float32 [3,H,W] in [0,1] (torchvision ToTensor, dataset.py:59-60)."""
    return patch_batch([image.contiguous()], [0], [0], image.shape[0], image.shape[1], PATCH_UNIT)[0]


def get_image_pair(dataset_config, idx, device=None):
    """dataset.py:9-62.  Returns (LR [3,h,w], HR [3,H,W], filename): float32 device tensors in [0,1]."""
    from PIL import Image
    dev = _device(device if device is not None else getattr(dataset_config, "device", None))
    hr_path = os.path.join(dataset_config.HR_dir, dataset_config.HR_images[idx])
    filename, _ = os.path.splitext(dataset_config.HR_images[idx])
    lr_path = os.path.join(dataset_config.LR_dir, f"{filename}x8.png")
    up = lambda path: torch.from_numpy(np.array(Image.open(path).convert("RGB"))).to(dev)      # :12,19
    lr_u8, hr_u8 = _shrink_pair(up(lr_path), up(hr_path), dataset_config.scale_factor, dataset_config.downsample)
    nt = dataset_config.noise_type
    if nt is not None:                                                                         # :50-55
        if nt["type"] == "SaltAndPepper":
            lr_u8 = degradation.add_salt_pepper_noise(lr_u8, s=nt["s"], p=nt["p"])
        elif nt["type"] == "Gaussian":
            lr_u8 = degradation.add_gaussian_noise(lr_u8, std=nt["std"])
    return to_tensor(lr_u8), to_tensor(hr_u8), filename


def _shrink_pair(lr_u8, hr_u8, scale_factor, extra_downsample):
    """dataset.py:21-45 on device uint8 images: both halved, LR optionally halved again, HR resized to scale_factor x LR (the
    `and` of :35 is kept as written)."""
    lr_u8 = degradation.downsample(lr_u8, 2)
    hr_u8 = degradation.downsample(hr_u8, 2)
    if extra_downsample:
        lr_u8 = degradation.downsample(lr_u8)
    h_lr, w_lr = lr_u8.shape[0], lr_u8.shape[1]
    w_hr, h_hr = scale_factor * w_lr, scale_factor * h_lr
    if w_hr > hr_u8.shape[1] and h_hr > hr_u8.shape[0]:
        w_hr = (hr_u8.shape[1] // scale_factor) * scale_factor
        h_hr = (hr_u8.shape[0] // scale_factor) * scale_factor
        w_lr, h_lr = w_hr // scale_factor, h_hr // scale_factor
        hr_u8 = degradation.resize(hr_u8, w_hr, h_hr)
        lr_u8 = degradation.resize(lr_u8, w_lr, h_lr)
    else:
        hr_u8 = degradation.resize(hr_u8, w_hr, h_hr)
    return lr_u8, hr_u8


class DIV2KDataset(torch.utils.data.Dataset):
    """dataset.py:68-95."""

    def __init__(self, LR_dir, scale_factor, downsample=False, noise_type=None, num_images=-1, HR_dir=None, device=None):
        super().__init__()
        self.downsample, self.noise_type, self.scale_factor = downsample, noise_type, scale_factor
        self.LR_dir, self.HR_dir, self.device = LR_dir, HR_dir, device
        self.LR_images, self.HR_images = os.listdir(LR_dir), os.listdir(HR_dir)
        if num_images > 0:
            self.LR_images, self.HR_images = self.LR_images[:num_images], self.HR_images[:num_images]

    def __getitem__(self, idx):
        return get_image_pair(self, idx)

    def __len__(self):
        return len(self.LR_images)


class GANDIV2KDataset(torch.utils.data.Dataset):
    """dataset.py:98-171."""

    def __init__(self, LR_dir, scale_factor, downsample=False, noise_type=None, num_images=-1, HR_dir=None, LR_patch_size=None,
                 train=False, device=None):
        super().__init__()
        self.train = train
        self.downsample, self.noise_type, self.scale_factor = downsample, noise_type, scale_factor
        self.LR_dir, self.HR_dir, self.device = LR_dir, HR_dir, device
        self.LR_images, self.HR_images = os.listdir(LR_dir), os.listdir(HR_dir)
        if num_images > 0:
            self.LR_images, self.HR_images = self.LR_images[:num_images], self.HR_images[:num_images]
        self.LR_patch_size = LR_patch_size

    def get_train_patches(self, LR_image, HR_image):
        """dataset.py:121-147 (the two randint draws from numpy's global generator, x first): views into the CHW tensors."""
        _, lr_h, lr_w = LR_image.size()
        top, left, hr_top, hr_left = train_patch_coords(lr_h, lr_w, self.LR_patch_size, self.scale_factor)
        pw, ph = self.LR_patch_size
        s = self.scale_factor
        return (LR_image[:, top:top + ph, left:left + pw], HR_image[:, hr_top:hr_top + ph * s, hr_left:hr_left + pw * s])

    @staticmethod
    def scale_images(LR_image, HR_image):
        """dataset.py:149-159 as written (in place): LR /= 255; HR = HR / 255 * 2 - 1, on tensors ToTensor already put in [0,1].
        On the HIP kernel dsr_scale_images_f32: ATen's device `x /= 255.0` multiplies by the reciprocal, one ulp off the host."""
        for t, mode in ((LR_image, PATCH_LR_REF), (HR_image, PATCH_HR_REF)):
            if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
                raise TypeError("scale_images: contiguous float32 device tensors expected")
            check(_lib.lib().dsr_scale_images_f32(_ptr(t), t.numel(), mode, _stream()))
        return LR_image, HR_image

    def __getitem__(self, idx):
        LR_image, HR_image, filename = get_image_pair(self, idx)
        LR_image, HR_image = GANDIV2KDataset.scale_images(LR_image, HR_image)
        if self.train:
            LR_image, HR_image = self.get_train_patches(LR_image, HR_image)
        return LR_image, HR_image, filename

    def __len__(self):
        return len(self.LR_images)


def train_patch_coords(lr_h, lr_w, LR_patch_size, scale_factor, rng=None):
    """(LR top, LR left, HR top, HR left) of dataset.py:121-141; `rng` defaults to numpy's global generator like the reference."""
    rng = np.random if rng is None else rng
    pw, ph = LR_patch_size
    cx = rng.randint(pw // 2, lr_w - pw // 2)
    cy = rng.randint(ph // 2, lr_h - ph // 2)
    left, top = int(cx - pw // 2), int(cy - ph // 2)
    return top, left, top * scale_factor, left * scale_factor


class PatchBank:
    """A (pre-shrunk, optionally degraded) image set resident in HBM as uint8, and batches of training patches cut from it on
    the device: ``sample(batch)`` = `batch` draws of (image index, patch position) + two launches of dsr_patch_batch_u8.

    pairs: iterable of (LR uint8 [h,w,3], HR uint8 [h*s, w*s, 3]) device tensors (e.g. from `_shrink_pair` + degradations)."""

    def __init__(self, pairs, scale_factor, LR_patch_size, reference_scaling=True, rng=None):
        self.lr = [p[0].contiguous() for p in pairs]
        self.hr = [p[1].contiguous() for p in pairs]
        if not self.lr:
            raise ValueError("PatchBank needs at least one image pair")
        for a, b in zip(self.lr, self.hr):
            if b.shape[0] < a.shape[0] * scale_factor or b.shape[1] < a.shape[1] * scale_factor:
                raise ValueError("an HR image is smaller than scale_factor x its LR image")
        self.scale, self.patch = scale_factor, tuple(LR_patch_size)
        self.modes = (PATCH_LR_REF, PATCH_HR_REF) if reference_scaling else (PATCH_UNIT, PATCH_HR_UNIT)
        self.rng = np.random if rng is None else rng

    def sample(self, batch, indices=None):
        """(LR [B,3,ph,pw], HR [B,3,ph*s,pw*s]) fp32 device batches; image b is `indices[b]` (default: uniform draws)."""
        if indices is None:
            indices = [int(self.rng.randint(0, len(self.lr))) for _ in range(batch)]
        pw, ph = self.patch
        tops, lefts, htops, hlefts = [], [], [], []
        for i in indices:
            t, l, ht, hl = train_patch_coords(self.lr[i].shape[0], self.lr[i].shape[1], self.patch, self.scale, self.rng)
            tops.append(t), lefts.append(l), htops.append(ht), hlefts.append(hl)
        lr = patch_batch([self.lr[i] for i in indices], tops, lefts, ph, pw, self.modes[0])
        hr = patch_batch([self.hr[i] for i in indices], htops, hlefts, ph * self.scale, pw * self.scale, self.modes[1])
        return lr, hr
