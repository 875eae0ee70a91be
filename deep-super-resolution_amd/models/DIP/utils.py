"""Building blocks of the Deep-Image-Prior skip network with the reference's surface
(/root/reference/models/DIP/utils.py): Concat (:10-41), act (:62-76), bn (:79-80), conv (:83-105).

Module objects are structure + parameter holders (state_dict keys match the reference, including the
1-indexed child names that its ``nn.Module.add`` monkey-patch produces); the arithmetic is done by
``run_fused`` below, which walks a Sequential and dispatches fused HIP kernels:
    [ReflectionPad2d] + Conv2d + BatchNorm2d + LeakyReLU  -> one ConvBNAct (padding folded into the tile loader)
    Concat(skip, deeper)                                  -> both branches + channel box-copy with centre crop
    BatchNorm2d on its own                                -> channel-stats + BN-apply kernels
    Upsample(scale 2, bilinear | nearest)                 -> bilinear2x / nearest2x kernel
    Conv2d + AvgPool2d|MaxPool2d (downsample_mode)        -> conv, 2x2 pool kernel, then BN(+act) on the pooled map
    Conv2d + Sigmoid at the very end                      -> conv with sigmoid epilogue, fp32 NCHW output
The reference's global monkey-patch of ``torch.nn.Module.add`` is not reproduced; ``add`` below is local.
"""
import torch
import torch.nn as nn

from ... import functional as F


def add(seq, module):
    """models/DIP/utils.py:5-8: children are named "1", "2", ..."""
    seq.add_module(str(len(seq) + 1), module)


class Concat(nn.Module):
    def __init__(self, dim, *args):
        super(Concat, self).__init__()
        self.dim = dim
        for idx, module in enumerate(args):
            self.add_module(str(idx), module)

    def __len__(self):
        return len(self._modules)

    def forward(self, input):
        raise RuntimeError("Concat is executed by run_fused (models/DIP/utils.py of this package)")


class GenNoise(nn.Module):
    """Unused by skip() (the call site is commented out in the reference, skip.py:58); kept for the surface."""

    def __init__(self, dim2):
        super(GenNoise, self).__init__()
        self.dim2 = dim2


def act(act_fun='LeakyReLU'):
    if isinstance(act_fun, str):
        if act_fun == 'LeakyReLU':
            return nn.LeakyReLU(0.2, inplace=True)
        elif act_fun == 'ELU':
            return nn.ELU()
        elif act_fun == 'none':
            return nn.Sequential()
        else:
            assert False
    else:
        return act_fun()


def bn(num_features):
    return nn.BatchNorm2d(num_features)


def conv(in_f, out_f, kernel_size, stride=1, bias=True, pad='zero', downsample_mode='stride'):
    downsampler = None
    if stride != 1 and downsample_mode != 'stride':
        if downsample_mode == 'avg':
            downsampler = nn.AvgPool2d(stride, stride)
        elif downsample_mode == 'max':
            downsampler = nn.MaxPool2d(stride, stride)
        else:
            assert False
        stride = 1
    to_pad = int((kernel_size - 1) / 2)
    padder = None
    if pad == 'reflection':
        padder = nn.ReflectionPad2d(to_pad)
        to_pad = 0
    convolver = nn.Conv2d(in_f, out_f, kernel_size, stride, padding=to_pad, bias=bias)
    return nn.Sequential(*[m for m in (padder, convolver, downsampler) if m is not None])


# ----------------------------------------------------------------------------- fused executor
def _conv_parts(seq):
    """A conv() Sequential -> (Conv2d, pad, pad_mode, pool) or None if `seq` is not one.
    pool: None | 'avg' | 'max' -- the 2x2 pooling that conv(..., downsample_mode=...) appends (:86-94, :104)."""
    if not isinstance(seq, nn.Sequential) or len(seq) == 0:
        return None
    mods = list(seq.children())
    pad_mode, pad = F.PAD_ZERO, 0
    if isinstance(mods[0], nn.ReflectionPad2d):
        pad_mode, pad = F.PAD_REFLECT, int(mods[0].padding[0])
        mods = mods[1:]
    if not mods or not isinstance(mods[0], nn.Conv2d) or len(mods) > 2:
        return None
    pool = None
    if len(mods) == 2:
        p = mods[1]
        if isinstance(p, nn.AvgPool2d):
            pool = 'avg'
        elif isinstance(p, nn.MaxPool2d):
            pool = 'max'
        else:
            return None

        def _two(v):
            return all(int(t) == 2 for t in (v if isinstance(v, (tuple, list)) else (v, v)))
        if not (_two(p.kernel_size) and _two(p.stride)):
            raise NotImplementedError("only 2x2 / stride-2 pooling after a conv is on the HIP path (skip() never builds another)")
    c = mods[0]
    if pad_mode == F.PAD_ZERO:
        pad = int(c.padding[0])
    elif pad == 0:
        pad_mode = F.PAD_ZERO
    return c, pad, pad_mode, pool


def _act_code(m):
    """act() module (:62-76) -> (activation code, slope) or None if `m` is not an activation with a kernel."""
    if isinstance(m, nn.LeakyReLU):
        return F.ACT_LEAKY, float(m.negative_slope)
    if isinstance(m, nn.ELU):
        if float(m.alpha) != 1.0:
            raise NotImplementedError("nn.ELU with alpha != 1 is not on the HIP path")
        return F.ACT_ELU, 0.0
    if isinstance(m, nn.Sequential) and len(m) == 0:       # act('none')
        return F.ACT_NONE, 0.0
    if isinstance(m, nn.ReLU):                             # act_fun given as a class (:75-76)
        return F.ACT_RELU, 0.0
    if isinstance(m, nn.Tanh):
        return F.ACT_TANH, 0.0
    return None


def run_fused(seq, x, c, train):
    """Execute an nn.Sequential built by skip() on an NHWC 16-bit tensor x with c real channels."""
    mods = list(seq.children())
    i = 0
    while i < len(mods):
        m = mods[i]
        cp = _conv_parts(m)
        if cp is not None:
            cv, pad, pmode, pool = cp
            geom = dict(stride=int(cv.stride[0]), pad=pad, pad_mode=pmode)
            nxt = mods[i + 1] if i + 1 < len(mods) else None
            nxt2 = mods[i + 2] if i + 2 < len(mods) else None
            if isinstance(nxt, nn.BatchNorm2d) and pool is None:
                a, slope, step = F.ACT_NONE, 0.0, 2
                ac = _act_code(nxt2) if nxt2 is not None else None
                if ac is not None:
                    (a, slope), step = ac, 3
                x = F.ConvBNAct.apply(x, cv.weight, cv.bias, nxt.weight, nxt.bias, nxt.running_mean, nxt.running_var,
                                      nxt.num_batches_tracked, None, None, dict(geom, act=a, slope=slope, train=train))
                c = cv.out_channels
                i += step
                continue
            if isinstance(nxt, nn.Sigmoid) and i + 2 == len(mods) and pool is None:
                return F.ConvOutNCHW.apply(x, cv.weight, cv.bias, dict(geom, act=F.ACT_SIGMOID)), cv.out_channels
            x = F.ConvAct.apply(x, cv.weight, cv.bias, None, dict(geom, act=F.ACT_NONE))
            if pool is not None:      # the statistics of a following BatchNorm are those of the POOLED map: no conv fusion
                x = (F.AvgPool2 if pool == 'avg' else F.MaxPool2).apply(x)
            c = cv.out_channels
            i += 1
            continue
        if isinstance(m, Concat):
            outs = [run_fused(b, x, c, train) for b in m.children()]
            assert len(outs) == 2, "skip() builds two-branch Concats"
            (ta, ca), (tb, cb) = outs
            x, c = F.ConcatCrop.apply(ta, tb, ca, cb), ca + cb
        elif isinstance(m, nn.BatchNorm2d):
            a, slope, step = F.ACT_NONE, 0.0, 1
            ac = _act_code(mods[i + 1]) if i + 1 < len(mods) else None
            if ac is not None:
                (a, slope), step = ac, 2
            x = F.BNAct.apply(x, m.weight, m.bias, m.running_mean, m.running_var, m.num_batches_tracked, c,
                              dict(act=a, slope=slope, train=train))
            i += step
            continue
        elif isinstance(m, nn.Upsample):
            if float(m.scale_factor) != 2.0 or m.mode not in ('bilinear', 'nearest') or m.align_corners:
                raise NotImplementedError("only Upsample(scale_factor=2, mode='bilinear'|'nearest') is on the HIP path")
            x = (F.Bilinear2x if m.mode == 'bilinear' else F.Nearest2x).apply(x)
        elif isinstance(m, nn.Sequential):
            x, c = run_fused(m, x, c, train)
        elif isinstance(m, nn.Sigmoid):
            raise NotImplementedError("Sigmoid must follow the last conv (need_sigmoid=True layout of skip())")
        else:
            raise NotImplementedError(f"module {type(m).__name__} has no HIP kernel on this path")
        i += 1
    return x, c
