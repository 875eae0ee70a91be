"""SRGAN generator with the reference's nn.Module surface, computed by the HIP kernels.

Mirror of /root/reference/models/GAN/generator.py (ResidualBlock :4-25, PixelShuffleBlock :27-41,
Generator :44-81): same class names, constructor arguments, child-module names and therefore the
same ``state_dict`` keys/shapes and default initialisation (children are created in the
reference's order, so a seeded construction draws identical parameters).  The child nn.Conv2d /
nn.BatchNorm2d / nn.PReLU objects are parameter holders only: ``forward`` never calls them, it
runs the fused conv(+BN)(+activation)(+residual / pixel-shuffle) HIP kernels.

Extension over the reference: ``factor`` may be any power of two >= 2 (log2(factor) shuffle
blocks); the reference accepts 8 and 16 only (:55-58) and raises UnboundLocalError otherwise.
"""
import math

import torch
import torch.nn as nn

from ... import functional as F


def _is_internal(x):
    return x.dtype in (torch.bfloat16, torch.float16)


class ResidualBlock(nn.Module):
    def __init__(self):
        super(ResidualBlock, self).__init__()
        self.conv1 = nn.Conv2d(in_channels=64, out_channels=64, kernel_size=3, stride=1, padding=1)
        self.bn1 = nn.BatchNorm2d(num_features=64)
        self.prelu1 = nn.PReLU()

        self.conv2 = nn.Conv2d(in_channels=64, out_channels=64, kernel_size=3, stride=1, padding=1)
        self.bn2 = nn.BatchNorm2d(num_features=64)
        self.compute_dtype = torch.bfloat16

    def _block(self, x, bn_updates=1):
        # carry_input: conv1's node hands the block input back for the skip connection, so that the two gradients of x
        # (conv path, skip path) meet inside its input-gradient launch instead of in an elementwise add
        cfg1 = dict(stride=1, pad=1, act=F.ACT_PRELU, train=self.training, bn_updates=bn_updates, carry_input=True)
        z, skip = F.ConvBNAct.apply(x, self.conv1.weight, self.conv1.bias, self.bn1.weight, self.bn1.bias,
                                    self.bn1.running_mean, self.bn1.running_var, self.bn1.num_batches_tracked,
                                    self.prelu1.weight, None, cfg1)                   # generator.py:15-18
        cfg2 = dict(stride=1, pad=1, act=F.ACT_NONE, train=self.training, bn_updates=bn_updates)
        return F.ConvBNAct.apply(z, self.conv2.weight, self.conv2.bias, self.bn2.weight, self.bn2.bias,
                                 self.bn2.running_mean, self.bn2.running_var, self.bn2.num_batches_tracked,
                                 None, skip, cfg2)                                    # :20-23 (x + z fused)

    def forward(self, x):
        if _is_internal(x):
            return self._block(x)
        c = x.shape[1]
        return F.ToNCHW.apply(self._block(F.ToNHWC.apply(x, self.compute_dtype)), c)


class PixelShuffleBlock(nn.Module):
    def __init__(self, in_channels):
        super(PixelShuffleBlock, self).__init__()
        self.conv1 = nn.Conv2d(in_channels=in_channels, out_channels=256, kernel_size=3, stride=1, padding=1)
        self.shuffler1 = nn.PixelShuffle(upscale_factor=2)
        self.prelu1 = nn.PReLU()
        self.compute_dtype = torch.bfloat16

    def _block(self, x, out_ps_link=None):
        cfg = dict(stride=1, pad=1, act=F.ACT_PRELU, pixel_shuffle=True)
        if out_ps_link is not None:
            cfg["out_ps_link"] = out_ps_link
        return F.ConvAct.apply(x, self.conv1.weight, self.conv1.bias, self.prelu1.weight, cfg)   # :37-39 fused

    def forward(self, x):
        if _is_internal(x):
            return self._block(x)
        return F.ToNCHW.apply(self._block(F.ToNHWC.apply(x, self.compute_dtype)), 64)


class Generator(nn.Module):
    def __init__(self, factor=8, residual_blocks_count=16):
        super(Generator, self).__init__()
        self.conv1 = nn.Conv2d(in_channels=3, out_channels=64, kernel_size=9, stride=1, padding=4)
        self.prelu1 = nn.PReLU()

        self.residual_blocks = nn.Sequential(*[ResidualBlock() for _ in range(residual_blocks_count)])

        self.conv2 = nn.Conv2d(in_channels=64, out_channels=64, kernel_size=3, stride=1, padding=1)
        self.bn1 = nn.BatchNorm2d(num_features=64)

        pixel_shuffles = int(round(math.log2(factor)))
        if factor < 2 or 2 ** pixel_shuffles != factor:
            raise ValueError(f"factor must be a power of two >= 2, got {factor}")

        self.pixel_shuffle_blocks = nn.Sequential(*[PixelShuffleBlock(in_channels=64) for _ in range(pixel_shuffles)])

        self.conv3 = nn.Conv2d(in_channels=64, out_channels=3, kernel_size=9, stride=1, padding=4)

        self.out = nn.Tanh()
        self.compute_dtype = torch.bfloat16     # float16 is accepted too (inference, BASELINE config 5)
        # momentum updates of the BatchNorm running statistics applied per train-mode forward.  The GAN step
        # recipe calls the generator twice on the same batch with the same weights (train_GAN.py:46,56): the two
        # outputs are identical, so steps.gan_step runs it once and sets this to 2.
        self.bn_updates = 1

    def forward(self, x):
        xi = F.ToNHWC.apply(x, self.compute_dtype)
        x0 = F.ConvAct.apply(xi, self.conv1.weight, self.conv1.bias, self.prelu1.weight,
                             dict(stride=1, pad=4, act=F.ACT_PRELU))                           # :68-69
        z = x0
        for block in self.residual_blocks:                                                        # :70
            z = block._block(z, self.bn_updates)
        z = F.ConvBNAct.apply(z, self.conv2.weight, self.conv2.bias, self.bn1.weight, self.bn1.bias,
                              self.bn1.running_mean, self.bn1.running_var, self.bn1.num_batches_tracked,
                              None, x0, dict(stride=1, pad=1, act=F.ACT_NONE, train=self.training,
                                       bn_updates=self.bn_updates))                                   # :71-74
        # (`psl`: the 9x9 tail's input-gradient launch also runs the backward of the last PixelShuffle + PReLU, dsr_conv_dgrad_ps:
        #  that block leaves its PReLU weight in the link and finds the masked, un-shuffled gradient of its conv output there)
        psl = {}
        nps = len(self.pixel_shuffle_blocks)
        for i, block in enumerate(self.pixel_shuffle_blocks):                                     # :76
            z = block._block(z, psl if i + 1 == nps else None)
        return F.ConvOutNCHW.apply(z, self.conv3.weight, self.conv3.bias,
                                   dict(stride=1, pad=4, act=F.ACT_TANH, in_ps_link=psl))       # :78-80
