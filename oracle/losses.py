"""Oracle (test infrastructure): loss arithmetic of the GAN step.

Follows /root/reference/utils/GAN.py:
  :96-98    get_adversarial_loss  = BCE(fake, 1)
  :101-105  get_loss_D            = BCE(real, 1) + BCE(fake, 0)
  :113-124  PerceptualLoss        = vgg_mse + adversarial (unweighted sum, :122)
nn.BCELoss: mean reduction, each log term clamped to >= -100 (torch semantics).
PSNR follows torchmetrics' definition 10*log10(range^2 / MSE); torchmetrics is absent
here, so that formula is "parity unpinned"  -- data range fixed to 2.0
for [-1, 1] tensors.
"""
import torch


def bce(p, target_value):
    """nn.BCELoss()(p, full_like(p, target_value)) for target in {0, 1}."""
    if target_value == 1:
        return -torch.clamp(torch.log(p), min=-100.0).mean()
    return -torch.clamp(torch.log(1.0 - p), min=-100.0).mean()


def loss_d(real_out, fake_out):
    return bce(real_out, 1) + bce(fake_out, 0)


def adversarial(fake_out):
    return bce(fake_out, 1)


def mse(a, b):
    return ((a - b) ** 2).mean()


def l1(a, b):
    return (a - b).abs().mean()


def psnr(a, b, data_range=2.0):
    m = ((a.double() - b.double()) ** 2).mean()
    return float(10.0 * torch.log10(torch.tensor(data_range, dtype=torch.float64) ** 2 / m))
