"""SRGAN discriminator with the reference's nn.Module surface, computed by the HIP kernels.

Mirror of /root/reference/models/GAN/discriminator.py (DiscriminatorConvBlock :4-19, Discriminator
:21-74): same constructor, child names (conv, convblocks.{i}.{conv1,bn1}, dense1, dense2), parameter
shapes -- dense1.weight is [1024, 512*H/16*W/16] in the reference's C,H,W flatten order -- and default
initialisation order.  ``fc_input_shape`` (:48-56) is computed arithmetically instead of by a dry
forward through torch (same number; a dry run would need the GPU at construction time).
"""
import torch
import torch.nn as nn

from ... import functional as F


class DiscriminatorConvBlock(nn.Module):
    def __init__(self, in_channels, out_channels, stride):
        super(DiscriminatorConvBlock, self).__init__()
        self.conv1 = nn.Conv2d(in_channels=in_channels, out_channels=out_channels, kernel_size=3, stride=stride, padding=1)
        self.bn1 = nn.BatchNorm2d(num_features=out_channels)
        self.leakyrelu = nn.LeakyReLU(negative_slope=0.2)
        self.stride = stride
        self.compute_dtype = torch.bfloat16

    def _block(self, x, first2=None, bn_out_link=None, bn_in_link=None):
        cfg = dict(stride=self.stride, pad=1, act=F.ACT_LEAKY, slope=0.2, train=self.training)
        if first2 is not None:
            cfg["first2"] = first2
        if bn_out_link is not None:
            cfg["bn_out_link"] = bn_out_link
        if bn_in_link is not None:
            cfg["bn_in_link"] = bn_in_link
        return F.ConvBNAct.apply(x, self.conv1.weight, self.conv1.bias, self.bn1.weight, self.bn1.bias,
                                 self.bn1.running_mean, self.bn1.running_var, self.bn1.num_batches_tracked,
                                 None, None, cfg)                                           # discriminator.py:15-17

    def forward(self, x):
        if x.dtype in (torch.bfloat16, torch.float16):
            return self._block(x)
        cout = self.conv1.out_channels
        return F.ToNCHW.apply(self._block(F.ToNHWC.apply(x, self.compute_dtype)), cout)


class Discriminator(nn.Module):
    def __init__(self, HR_image_shape):
        super(Discriminator, self).__init__()
        self.conv = nn.Conv2d(in_channels=3, out_channels=64, kernel_size=3, stride=1, padding=1)
        self.leakyrelu1 = nn.LeakyReLU(negative_slope=0.2)
        self.convblocks = nn.Sequential(*[DiscriminatorConvBlock(in_channels=64, out_channels=64, stride=2),
                                          DiscriminatorConvBlock(in_channels=64, out_channels=128, stride=1),
                                          DiscriminatorConvBlock(in_channels=128, out_channels=128, stride=2),
                                          DiscriminatorConvBlock(in_channels=128, out_channels=256, stride=1),
                                          DiscriminatorConvBlock(in_channels=256, out_channels=256, stride=2),
                                          DiscriminatorConvBlock(in_channels=256, out_channels=512, stride=1),
                                          DiscriminatorConvBlock(in_channels=512, out_channels=512, stride=2)])
        dense1_shape = self.fc_input_shape(HR_image_shape)
        self.dense1 = nn.Linear(in_features=dense1_shape, out_features=1024)
        self.dense1.weight._dsr_dense_head = True    # dist.GradSync: gather this gradient's factors instead of reducing it
        self.leakyrelu2 = nn.LeakyReLU(negative_slope=0.2)
        self.dense2 = nn.Linear(1024, 1)
        self.sigmoid = nn.Sigmoid()
        self.compute_dtype = torch.bfloat16

    def fc_input_shape(self, HR_image_shape):
        h, w = int(HR_image_shape[0]), int(HR_image_shape[1])
        for blk in self.convblocks:
            s = blk.stride
            h, w = (h + 2 - 3) // s + 1, (w + 2 - 3) // s + 1
        return 512 * h * w

    def features(self, x):
        """conv stack (:60-63) -> NHWC 16-bit [N, H/16, W/16, 512]"""
        xi = F.ToNHWC.apply(x, self.compute_dtype)
        blk0 = self.convblocks[0]
        fused = blk0.training and F.first2_supported(xi, self.conv.weight, blk0.conv1.weight, blk0.stride)
        # fused: the first layer is computed INSIDE the second layer's forward kernel (its 64-channel activation, 1.07 GB at
        # 512x512 x 32, is recomputed per tile and written only for a backward pass); ConvAct then only allocates and records
        # (`link`: where the fused backward of the two layers, dsr_conv_dgrad_first_bwd, leaves the first layer's gradients for
        #  its autograd node -- a step that needs no image gradient never writes the gradient of the activation between them)
        link = {} if fused else None
        z = F.ConvAct.apply(xi, self.conv.weight, self.conv.bias, None,
                            dict(stride=1, pad=1, act=F.ACT_LEAKY, slope=0.2, defer=fused, first2_link=link))  # :60-61
        # (`bnl`: a stride-2 block's input-gradient launch also forms the BatchNorm-backward sums of the block in front of it,
        #  dsr_conv_dgrad_bn: that block leaves its raw conv output and affine map in the link, and finds the partial rows there)
        nb = len(self.convblocks)
        bnl = [{} if (i + 1 < nb and self.convblocks[i + 1].stride == 2) else None for i in range(nb)]
        for i, blk in enumerate(self.convblocks):                                             # :63
            z = blk._block(z, (xi, self.conv.weight, self.conv.bias, 0.2, link) if (fused and i == 0) else None,
                           bn_out_link=bnl[i], bn_in_link=bnl[i - 1] if i > 0 else None)
        return z

    def head(self, z):
        return F.DenseHead.apply(z, self.dense1.weight, self.dense1.bias, self.dense2.weight, self.dense2.bias,
                                 512)                                                          # :65-72

    def forward(self, x):
        return self.head(self.features(x))

    def forward_pair(self, a, b, fa=None):
        """(self(a), self(b)) with ONE pass over the dense head.

        The conv stack runs separately on each batch, so every train-mode BatchNorm sees exactly the statistics (and
        running-stat updates, in the same order) of two separate calls; the dense layers are per-sample, so running
        them on the concatenated batch is arithmetically the same while dense1's K x 1024 weight (2.1 GB at 512x512)
        is streamed once instead of twice in forward, dgrad and wgrad, and its two gradient contributions are summed
        inside the MFMA contraction instead of by a 6 GB elementwise add."""
        if fa is None:                   # (`fa`: features(a) computed earlier by the caller, e.g. beside the generator forward)
            fa = self.features(a)
        fb = self.features(b)
        if fa.shape[0] + fb.shape[0] > 64:
            return self.head(fa), self.head(fb)
        out = self.head(torch.cat([fa, fb], dim=0))
        return out[:fa.shape[0]], out[fa.shape[0]:]
