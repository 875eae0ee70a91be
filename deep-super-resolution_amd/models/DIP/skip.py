"""skip(): the DIP encoder-decoder with skip connections (mirror of /root/reference/models/DIP/skip.py:3-96).

Same arguments, same nesting and the same 1-indexed child names, so ``state_dict()`` keys are identical
(e.g. ``1.0.1.1.weight``, ``1.1.7.3.1.weight``, ``9.1.weight``).  Returns a SkipNet (an nn.Sequential whose
forward is the fused HIP executor)."""
import torch
import torch.nn as nn

from ... import functional as F
from .utils import Concat, act, add, bn, conv, run_fused


class SkipNet(nn.Sequential):
    # fp16 storage: the hourglass normalises tiny populations (2x2..8x8 maps at batch 1), which amplifies storage
    # rounding; fp16's 11-bit mantissa keeps the net within ~1e-2 of fp32 where bf16's 8 bits drift to ~1e-1.
    # BatchNorm keeps every activation O(1), so fp16's range is not an issue; gradients use a static loss scale.
    compute_dtype = torch.float16

    def forward(self, x):
        out, c_out = run_fused(self, F.ToNHWC.apply(x, self.compute_dtype), x.shape[1], self.training)
        if out.dtype != torch.float32:      # need_sigmoid=False: hand back fp32 NCHW like the reference would
            out = F.ToNCHW.apply(out, c_out)
        return out


def _per_scale(v, n):
    return list(v) if isinstance(v, (list, tuple)) else [v] * n


def skip(num_input_channels=2, num_output_channels=3,
         num_channels_down=[16, 32, 64, 128, 128], num_channels_up=[16, 32, 64, 128, 128],
         num_channels_skip=[4, 4, 4, 4, 4], filter_size_down=3, filter_size_up=3, filter_skip_size=1,
         need_sigmoid=True, need_bias=True, pad='zero', upsample_mode='nearest', downsample_mode='stride',
         act_fun='LeakyReLU', need1x1_up=True):
    assert len(num_channels_down) == len(num_channels_up) == len(num_channels_skip)
    n_scales = len(num_channels_down)
    upsample_mode = _per_scale(upsample_mode, n_scales)
    downsample_mode = _per_scale(downsample_mode, n_scales)
    filter_size_down = _per_scale(filter_size_down, n_scales)
    filter_size_up = _per_scale(filter_size_up, n_scales)
    last_scale = n_scales - 1

    model = SkipNet()
    level = model                 # the Sequential currently being filled (reference: model_tmp)
    depth_in = num_input_channels
    for i in range(n_scales):
        deeper, skip_branch = nn.Sequential(), nn.Sequential()
        has_skip = num_channels_skip[i] != 0
        add(level, Concat(1, skip_branch, deeper) if has_skip else deeper)
        k = num_channels_up[i + 1] if i < last_scale else num_channels_down[i]
        add(level, bn(num_channels_skip[i] + k))

        if has_skip:
            add(skip_branch, conv(depth_in, num_channels_skip[i], filter_skip_size, bias=need_bias, pad=pad))
            add(skip_branch, bn(num_channels_skip[i]))
            add(skip_branch, act(act_fun))

        add(deeper, conv(depth_in, num_channels_down[i], filter_size_down[i], 2, bias=need_bias, pad=pad,
                         downsample_mode=downsample_mode[i]))
        add(deeper, bn(num_channels_down[i]))
        add(deeper, act(act_fun))
        add(deeper, conv(num_channels_down[i], num_channels_down[i], filter_size_down[i], bias=need_bias, pad=pad))
        add(deeper, bn(num_channels_down[i]))
        add(deeper, act(act_fun))

        deeper_main = nn.Sequential()
        if i != last_scale:
            add(deeper, deeper_main)
        add(deeper, nn.Upsample(scale_factor=2, mode=upsample_mode[i]))

        add(level, conv(num_channels_skip[i] + k, num_channels_up[i], filter_size_up[i], 1, bias=need_bias, pad=pad))
        add(level, bn(num_channels_up[i]))
        add(level, act(act_fun))
        if need1x1_up:
            add(level, conv(num_channels_up[i], num_channels_up[i], 1, bias=need_bias, pad=pad))
            add(level, bn(num_channels_up[i]))
            add(level, act(act_fun))

        depth_in = num_channels_down[i]
        level = deeper_main

    add(model, conv(num_channels_up[0], num_output_channels, 1, bias=need_bias, pad=pad))
    if need_sigmoid:
        add(model, nn.Sigmoid())
    return model
