"""The timed step recipes on the HIP path.

gen_l1_step : BASELINE config 2 (generator-only x4, L1, Adam) -- defined by SURVEY.md 8(d).
gan_step    : train_GAN.py:38-71 (do_epoch): D step, then G step with the detached adversarial term.
dip_step    : DIP.py:47-95 closure + utils/DIP.py:33-40 Adam iteration.
"""
import torch

from . import functional as F


def gen_l1_step(gen, opt, lr_patches, hr_patches):
    fake = gen(lr_patches)
    loss = F.l1_loss(fake, hr_patches)
    opt.zero_grad()
    loss.backward()
    opt.step()
    return loss.detach(), fake.detach()
