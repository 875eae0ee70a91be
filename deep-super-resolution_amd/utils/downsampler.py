"""Downsampler with the reference's surface (utils/downsampler.py:5-71), computed by a HIP kernel.

Same constructor arguments, assertions and attributes (``kernel`` float64 numpy, ``downsampler_``
nn.Conv2d holding the dense-diagonal weight so state_dicts/``.parameters()`` match, ``padding``,
``preserve_size``, ``x``).  The reference's Conv2d(n, n, k, stride=f) has only its n diagonal filters
non-zero and zero bias (:44-50), i.e. it is a depthwise correlation with one k x k kernel -- that is what
the device kernel computes (from ``downsampler_.weight[0, 0]``, the float32 copy the reference makes at :48-50).
get_kernel (:73-135) is restated vectorised in numpy; it is host-side construction-time work.
"""
import numpy as np
import torch
import torch.nn as nn

from .. import functional as F


class Downsampler(nn.Module):
    def __init__(self, n_planes, factor, kernel_type, phase=0, kernel_width=None, support=None, sigma=None,
                 preserve_size=False):
        super(Downsampler, self).__init__()

        assert phase in [0, 0.5], 'phase should be 0 or 0.5'

        # named presets -> (family, kernel_width, support, sigma); None keeps the caller's value
        presets = {
            'lanczos2': ('lanczos', 4 * factor + 1, 2, None),
            'lanczos3': ('lanczos', 6 * factor + 1, 3, None),
            'gauss12': ('gauss', 7, None, 1 / 2),
            'gauss1sq2': ('gauss', 9, None, 1. / np.sqrt(2)),
        }
        if kernel_type in presets:
            kernel_type_, kernel_width, sup_, sig_ = presets[kernel_type]
            support = sup_ if sup_ is not None else support
            sigma = sig_ if sig_ is not None else sigma
        else:
            assert kernel_type in ('lanczos', 'gauss', 'box'), 'wrong name kernel'
            kernel_type_ = kernel_type

        self.kernel = get_kernel(factor, kernel_type_, phase, kernel_width, support=support, sigma=sigma)

        downsampler = nn.Conv2d(n_planes, n_planes, kernel_size=self.kernel.shape, stride=factor, padding=0)
        downsampler.weight.data[:] = 0
        downsampler.bias.data[:] = 0
        kernel_torch = torch.from_numpy(self.kernel)
        for i in range(n_planes):
            downsampler.weight.data[i, i] = kernel_torch
        self.downsampler_ = downsampler

        self.pad = 0
        if preserve_size:
            if self.kernel.shape[0] % 2 == 1:
                pad = int((self.kernel.shape[0] - 1) / 2.)
            else:
                pad = int((self.kernel.shape[0] - factor) / 2.)
            self.padding = nn.ReplicationPad2d(pad)
            self.pad = pad

        self.preserve_size = preserve_size
        self.factor = factor

    def forward(self, input):
        self.x = input          # the reference caches the (padded) input here (:70); the pad is fused on device
        kern = self.downsampler_.weight[0, 0].detach().contiguous()
        return F.Downsample.apply(input, kern, self.factor, self.pad if self.preserve_size else 0)


def get_kernel(factor, kernel_type, phase, kernel_width, support=None, sigma=None):
    assert kernel_type in ['lanczos', 'gauss', 'box']

    if phase == 0.5 and kernel_type != 'box':
        size = kernel_width - 1
    else:
        size = kernel_width

    if kernel_type == 'box':
        assert phase == 0.5, 'Box filter is always half-phased'
        kernel = np.full([size, size], 1. / (kernel_width * kernel_width))
    elif kernel_type == 'gauss':
        assert sigma, 'sigma is not specified'
        assert phase != 0.5, 'phase 1/2 for gauss not implemented'
        center = (kernel_width + 1.) / 2.
        d = (np.arange(1, size + 1, dtype=np.float64) - center) / 2.
        sigma_sq = sigma * sigma
        kernel = np.exp(-(d[:, None] ** 2 + d[None, :] ** 2) / (2 * sigma_sq)) / (2. * np.pi * sigma_sq)
    else:
        assert support, 'support is not specified'
        center = (kernel_width + 1) / 2.
        idx = np.arange(1, size + 1, dtype=np.float64)
        d = np.abs(idx + 0.5 - center) / factor if phase == 0.5 else np.abs(idx - center) / factor
        with np.errstate(divide='ignore', invalid='ignore'):
            v = support * np.sin(np.pi * d) * np.sin(np.pi * d / support) / (np.pi * np.pi * d * d)
        v = np.where(d != 0, v, 1.0)
        kernel = v[:, None] * v[None, :]

    kernel /= kernel.sum()
    return kernel
