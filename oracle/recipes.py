"""Oracle (test infrastructure): the timed step recipes, restated on CPU fp32.

Follows /root/reference:
  train_GAN.py:35-36   Adam(lr) for G and for D (default betas/eps, no weight decay)
  train_GAN.py:38-71   do_epoch: D step then G step, incl. the ``.detach()`` at :58
  DIP.py:47-95         closure (input jitter, net, downsampler, MSE, backward)
  utils/DIP.py:33-40   optimize('adam')
train_GAN.py / DIP.py cannot be imported under Python 3.10 (PEP-701 f-strings), so these
follow the source text; the generator-only L1 step (BASELINE config 2) has no reference
counterpart and is defined by SURVEY.md 8(d).
"""
import torch

from . import dip, downsampler, gan, losses, vgg


def leaves(sd, keys=None):
    """Turn the trainable entries of a state_dict into autograd leaves (in place)."""
    keys = gan.trainable(sd) if keys is None else keys
    for k in keys:
        sd[k] = sd[k].detach().clone().requires_grad_(True)
    return [sd[k] for k in keys]


def zero_grad(params):
    for p in params:
        p.grad = None


class GanState:
    def __init__(self, g_sd, d_sd, vgg_sd, lr, vgg_resize=256, vgg_crop=224):
        self.g, self.d, self.vgg = g_sd, d_sd, vgg_sd
        self.g_params = leaves(self.g)
        self.d_params = leaves(self.d)
        self.opt_g = torch.optim.Adam(self.g_params, lr=lr)       # train_GAN.py:35
        self.opt_d = torch.optim.Adam(self.d_params, lr=lr)       # train_GAN.py:36
        self.vgg_resize, self.vgg_crop = vgg_resize, vgg_crop


def gan_step(st, lr_patches, hr_patches, capture=None):
    """One do_epoch (train_GAN.py:38-71).  Returns (loss_D, loss_G, fake) as floats/tensor.

    ``capture`` (a dict, tests only) receives clones of the gradients the two optimiser steps consume:
    ``capture["d_grads"]`` after :52 (before the dead BCE gradient of :63 is accumulated on top of them) and
    ``capture["g_grads"]`` after :63, keyed like the state_dicts."""
    real_d = gan.discriminator_forward(st.d, hr_patches, True)              # :44
    fake = gan.generator_forward(st.g, lr_patches, True).detach()           # :46
    fake_d = gan.discriminator_forward(st.d, fake, True)                    # :47
    loss_d = losses.loss_d(real_d, fake_d)                                  # :48
    zero_grad(st.d_params)                                                  # :51
    loss_d.backward()                                                       # :52
    if capture is not None:
        capture["d_grads"] = {k: st.d[k].grad.detach().clone() for k in gan.trainable(st.d)}
    st.opt_d.step()                                                         # :53

    fake = gan.generator_forward(st.g, lr_patches, True)                    # :56
    fake_d = gan.discriminator_forward(st.d, fake.detach(), True)           # :58 (detached!)
    content = vgg.vgg_loss(st.vgg, fake, hr_patches, st.vgg_resize, st.vgg_crop)
    loss_g = content + losses.adversarial(fake_d)                           # :59, utils/GAN.py:122
    zero_grad(st.g_params)                                                  # :62
    loss_g.backward()                                                       # :63
    if capture is not None:
        capture["g_grads"] = {k: st.g[k].grad.detach().clone() for k in gan.trainable(st.g)}
        capture["content"] = float(content.detach())
    st.opt_g.step()                                                         # :64
    return float(loss_d.detach()), float(loss_g.detach()), fake.detach()


class GenOnlyState:
    def __init__(self, g_sd, lr):
        self.g = g_sd
        self.g_params = leaves(self.g)
        self.opt_g = torch.optim.Adam(self.g_params, lr=lr)


def gen_l1_step(st, lr_patches, hr_patches):
    """BASELINE config 2: L1(G(LR), HR), Adam."""
    fake = gan.generator_forward(st.g, lr_patches, True)
    loss = losses.l1(fake, hr_patches)
    zero_grad(st.g_params)
    loss.backward()
    st.opt_g.step()
    return float(loss.detach()), fake.detach()


class DipState:
    def __init__(self, net_sd, cfg, net_input, factor, lr, reg_noise_std):
        self.net, self.cfg = net_sd, cfg
        self.params = leaves(self.net)                                # get_params('net') utils/DIP.py:57-58
        self.opt = torch.optim.Adam(self.params, lr=lr)               # utils/DIP.py:34
        self.net_input_saved = net_input.detach().clone()             # DIP.py:33
        self.noise = net_input.detach().clone()                       # DIP.py:34
        self.net_input = net_input
        self.factor, self.sigma = factor, reg_noise_std


def dip_step(st, lr_image, noise=None):
    """optimizer.zero_grad(); closure(); optimizer.step()  (utils/DIP.py:35-38, DIP.py:47-68).

    ``noise`` lets a test inject the N(0,1) draw (DIP.py:52 draws it from the global CPU
    generator with ``noise.normal_()``)."""
    zero_grad(st.params)
    if st.sigma > 0:
        if noise is None:
            noise = st.noise.normal_()
        st.net_input = st.net_input_saved + noise * st.sigma               # DIP.py:52
    out_hr = dip.skip_forward(st.net, st.net_input, st.cfg, True)          # :60
    out_lr = downsampler.downsampler_forward(out_hr, st.factor, "lanczos2", phase=0.5,
                                             preserve_size=True)           # :62, :29
    loss = losses.mse(out_lr, lr_image)                                    # :65
    loss.backward()                                                        # :68
    st.opt.step()
    return float(loss.detach()), out_hr.detach()
