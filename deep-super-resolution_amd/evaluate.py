"""Checkpoint interchange and the generator evaluation loop on the HIP path.

Counterparts of the reference's host glue, kept format-compatible so that ``.pth`` files move both ways:
  save_model / load_model      utils/common.py:11-18, 46-60  (``torch.save(model.state_dict())``; ``module.`` prefix of a
                               DataParallel/DDP-wrapped model stripped on load)
  evaluate_generator           eval_GAN.py:21-69  (``gan_G.eval()``, one image at a time, PSNR average, PNG dump) --
                               run under ``no_grad`` and optionally halo-tiled (infer.super_resolve), which the reference's
                               loop lacks (its images are pre-shrunk "because too big for the forward pass", dataset.py:21-23)

PSNR is 10*log10(range^2 / MSE) (torchmetrics' definition; that package is absent here, so the formula is restated --
"parity unpinned").  ``data_range=None`` infers max-min of the target like ``PeakSignalNoiseRatio()`` does.
The MSE is reduced on the device by the HIP loss kernel; only the final scalar crosses to the host.
"""
import math
import os
from collections import OrderedDict

import torch

from . import functional as F
from . import infer


def save_model(model, name, out_dir):
    """utils/common.py:11-18: <out_dir>/<name>.pth holding model.state_dict() (tensors moved to the host so that the file
    loads anywhere)."""
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, f"{name}.pth")
    torch.save(OrderedDict((k, v.detach().cpu()) for k, v in model.state_dict().items()), path)
    return path


def load_model(model, model_path):
    """utils/common.py:46-60: load a reference-format checkpoint, with or without the ``module.`` key prefix.
    ``weights_only=True`` (as in the reference): nothing in the file is executed."""
    state = torch.load(model_path, map_location="cpu", weights_only=True)
    clean = OrderedDict((k[len("module."):] if k.startswith("module.") else k, v) for k, v in state.items())
    model.load_state_dict(clean)
    return model


def psnr(pred, target, data_range=None):
    """Host float: 10*log10(range^2 / MSE(pred, target)) for fp32 NCHW device tensors.
    ``data_range=None`` follows what torchmetrics' ``PeakSignalNoiseRatio()`` (eval_GAN.py:30,47: the metric is CALLED per
    image, which evaluates that image from the metric's default state) does as far as its published behaviour goes: the range
    is max(target.max(), 0) - min(target.min(), 0) -- its running minimum and maximum start at 0, so a [0.2, 0.9] target has
    range 0.9, not 0.7.  torchmetrics is absent from /root/reference and from this image: PARITY UNPINNED."""
    mse = float(F.mse_loss(pred.detach(), target.detach()))
    if data_range is None:
        data_range = max(float(target.max()), 0.0) - min(float(target.min()), 0.0)
    return 10.0 * math.log10(data_range ** 2 / mse) if mse > 0 else float("inf")


def ssim(pred, target, data_range=1.0):
    """Host float: mean SSIM (Gaussian 11x11, sigma 1.5, K1 0.01, K2 0.03 -- the torchmetrics defaults the reference's scripts
    run with, ``SSIM(data_range=1.)`` at eval_GAN.py:31) of fp32 NCHW device tensors, on the HIP kernel dsr_ssim_f32."""
    import ctypes as C
    from . import _lib
    pred, target = pred.detach().contiguous().float(), target.detach().contiguous().float()
    if pred.shape != target.shape or pred.dim() != 4:
        raise RuntimeError(f"ssim: operands must be two [N,C,H,W] tensors of one shape, got {tuple(pred.shape)} and {tuple(target.shape)}")
    n, c, h, w = pred.shape
    lib = _lib.lib()
    blocks = lib.dsr_ssim_blocks(n * c, h, w)
    if blocks <= 0:
        raise RuntimeError(f"ssim: images of {h}x{w} are smaller than the 11x11 window")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rows = (blocks + 63) // 64                      # block sums as a [rows][64] table (zero padded): two-stage column sum
    part = torch.zeros(rows * 64, dtype=torch.float32, device=pred.device)
    _lib.check(lib.dsr_ssim_f32(C.c_void_p(pred.data_ptr()), C.c_void_p(target.data_ptr()), n * c, h, w, float(data_range),
                                C.c_void_p(part.data_ptr()), st))
    cols = torch.empty(64, dtype=torch.float32, device=pred.device)
    total = torch.empty(1, dtype=torch.float32, device=pred.device)
    _lib.check(lib.dsr_pw_sum_rows(C.c_void_p(part.data_ptr()), rows, 64, 0, 64, 1.0, C.c_void_p(cols.data_ptr()), 0, 1, st))
    _lib.check(lib.dsr_pw_sum_rows(C.c_void_p(cols.data_ptr()), 64, 1, 0, 1, 1.0 / (n * c * (h - 10) * (w - 10)),
                                   C.c_void_p(total.data_ptr()), 0, 0, st))
    return float(total)


def to_uint8_image(img):
    """[3,H,W] float in [0,1] -> [H,W,3] uint8 numpy (eval_GAN.py:55-56; values are clipped first, the reference's bare
    ``astype(np.uint8)`` wraps out-of-range values)."""
    return (img.detach().clamp(0.0, 1.0) * 255.0).round().to(torch.uint8).permute(1, 2, 0).cpu().numpy()


def evaluate_generator(gen, pairs, tile=None, out_dir=None, data_range=None, dtype=torch.float16, to_unit=None,
                       with_ssim=True):
    """eval_GAN.py:21-69 for an iterable of (LR [1,3,h,w], HR [1,3,H,W], name) on the device.

    Returns {'avg_psnr': ..., 'psnr': {name: value}, 'avg_ssim': ..., 'ssim': {name: value}} (LPIPS needs a downloaded
    AlexNet and is out of reach offline).  ``out_dir`` (optional) receives <out_dir>/images/<name>.png like
    save_image (utils/common.py:20-33); ``to_unit`` maps the network's output range to [0,1] for the PNG (default:
    identity, as in the reference)."""
    per, ssims = OrderedDict(), OrderedDict()
    for lr_image, hr_image, name in pairs:
        if isinstance(name, (list, tuple)):
            name = name[0]                                  # DataLoader collation of a batch of one (eval_GAN.py:40)
        resolved = infer.super_resolve(gen, lr_image, tile=tile, dtype=dtype)
        per[name] = psnr(resolved, hr_image, data_range)
        if with_ssim:
            ssims[name] = ssim(resolved, hr_image, 1.0)          # SSIM(data_range=1.) as at eval_GAN.py:31
        if out_dir is not None:
            from PIL import Image
            img_dir = os.path.join(out_dir, "images")
            os.makedirs(img_dir, exist_ok=True)
            img = resolved[0] if to_unit is None else to_unit(resolved[0])
            Image.fromarray(to_uint8_image(img)).save(os.path.join(img_dir, f"{name}.png"))
    out = {"avg_psnr": sum(per.values()) / max(len(per), 1), "psnr": per}
    if with_ssim:
        out.update(avg_ssim=sum(ssims.values()) / max(len(ssims), 1), ssim=ssims)
    return out
