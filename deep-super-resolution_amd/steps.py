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


def gan_step(gen, disc, perceptual, opt_g, opt_d, lr_patches, hr_patches, sync_g=None, sync_d=None):
    """train_GAN.py:38-71.  Returns (loss_D, loss_G, fake) as device tensors (no host sync here)."""
    # train_GAN.py:46 and :56 evaluate gen(lr_patches) twice with the same weights and batch statistics -- the
    # two outputs are bit-identical and only the BatchNorm running statistics notice the second call.  One forward
    # (with the autograd graph the G step needs) + a double running-stat update is exactly equivalent.
    gen.bn_updates = 2
    fake = gen(lr_patches)                                       # :46 and :56
    gen.bn_updates = 1
    # --- discriminator
    real_d, fake_d = disc.forward_pair(hr_patches, fake.detach())   # :44, :47 (BN statistics per batch, as there)
    loss_d = F.bce_const(real_d, 1.0) + F.bce_const(fake_d, 0.0)  # :48, utils/GAN.py:101-105
    opt_d.zero_grad()                                            # :51 (gan_D.zero_grad())
    loss_d.backward()                                            # :52
    if sync_d is not None:
        sync_d()
    opt_d.step()                                                 # :53
    # --- generator
    with torch.no_grad():
        # :58 detaches the generator output, so this D pass never sends a gradient anywhere that survives
        # (D's .grad from it is wiped by the next zero_grad, :51); it still updates D's BN running statistics.
        fake_d = disc(fake.detach())
    loss_g = perceptual(fake, hr_patches, fake_d, None)          # :59
    opt_g.zero_grad()                                            # :62
    loss_g.backward()                                            # :63
    if sync_g is not None:
        sync_g()
    opt_g.step()                                                 # :64
    return loss_d.detach(), loss_g.detach(), fake.detach()


class DipRunner:
    """DIP.py:22-123 state: fixed noise input, jitter buffer, Lanczos downsampler, Adam over the net."""

    def __init__(self, net, downsampler, net_input, lr_image, learning_rate, reg_noise_std, loss_scale=None):
        from .optim import FusedAdam
        # fp16 storage (the DIP default, see models/DIP/skip.py) needs a static loss scale so that activation
        # gradients (~1e-5 at the MSE) stay out of fp16's subnormal range; Adam un-scales on the fly.
        if loss_scale is None:
            loss_scale = 1024.0 if getattr(net, "compute_dtype", None) == torch.float16 else 1.0
        self.loss_scale = float(loss_scale)
        self.net, self.down = net, downsampler
        self.net_input_saved = net_input.detach().clone()        # DIP.py:33
        self.noise = net_input.detach().clone()                  # DIP.py:34
        self.net_input = net_input
        self.lr_image = lr_image
        self.sigma = reg_noise_std
        self.opt = FusedAdam(list(net.parameters()), lr=learning_rate,
                             grad_scale=1.0 / self.loss_scale)       # get_params('net') + utils/DIP.py:34

    def step(self, noise=None):
        """optimizer.zero_grad(); closure(); optimizer.step()  (utils/DIP.py:35-38, DIP.py:47-68)."""
        self.opt.zero_grad()
        if self.sigma > 0:
            if noise is None:
                noise = self.noise.normal_()                     # DIP.py:52 (torch RNG on whatever device holds it)
            self.net_input = self.net_input_saved + noise * self.sigma
        out_hr = self.net(self.net_input)                        # :60
        out_lr = self.down(out_hr)                               # :62
        loss = F.mse_loss(out_lr, self.lr_image)                 # :65
        (loss * self.loss_scale if self.loss_scale != 1.0 else loss).backward()   # :68
        self.opt.step()
        return loss.detach(), out_hr.detach()
