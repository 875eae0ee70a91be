"""Oracle (test infrastructure): fp32 CPU restatement of the Deep-Image-Prior "skip"
hourglass as a pure function over a reference-format ``state_dict``.

Follows /root/reference:
  models/DIP/__init__.py:8-18   get_net (only NET_TYPE == 'skip')
  models/DIP/skip.py:3-96       skip(): nested encoder-decoder, module names 1-indexed
  models/DIP/utils.py:5-8       nn.Module.add -> names "1","2",...
  models/DIP/utils.py:18-38     Concat with centre crop to the smallest H, W
  models/DIP/utils.py:62-76     act(): LeakyReLU(0.2) | ELU | none
  models/DIP/utils.py:83-105    conv(): [ReflectionPad2d(int((k-1)/2))] + Conv2d(k, stride, padding 0)
                                [+ AvgPool2d|MaxPool2d(stride, stride) after a stride-1 conv when downsample_mode != 'stride']
  utils/DIP.py:70-96            fill_noise / get_noise
"""
import torch
import torch.nn.functional as F

from .gan import _bn_shapes, batch_norm


class SkipConfig:
    """Arguments of get_net (models/DIP/__init__.py:8) reduced to what skip() consumes."""

    def __init__(self, input_depth=32, n_channels=3, skip_n33d=128, skip_n33u=128, skip_n11=4,
                 num_scales=5, pad="reflection", upsample_mode="bilinear", act_fun="LeakyReLU",
                 downsample_mode="stride"):
        self.input_depth = input_depth
        self.n_channels = n_channels
        self.down = [skip_n33d] * num_scales if isinstance(skip_n33d, int) else list(skip_n33d)
        self.up = [skip_n33u] * num_scales if isinstance(skip_n33u, int) else list(skip_n33u)
        self.skip = [skip_n11] * num_scales if isinstance(skip_n11, int) else list(skip_n11)
        assert len(self.down) == len(self.up) == len(self.skip)      # skip.py:19
        assert all(c != 0 for c in self.skip), "oracle restates the Concat branch only"
        self.pad = pad
        self.upsample_mode = upsample_mode
        assert act_fun in ("LeakyReLU", "ELU", "none") and downsample_mode in ("stride", "avg", "max")
        self.act_fun = act_fun
        self.downsample_mode = downsample_mode
        self.ci = 1 if pad == "reflection" else 0   # index of the Conv2d inside conv()'s Sequential


def skip_shapes(cfg):
    s = {}
    n = len(cfg.down)
    c = str(cfg.ci)

    def level(P, i, depth_in):
        last = i == n - 1
        s[P + "1.0.1." + c + ".weight"] = (cfg.skip[i], depth_in, 1, 1)        # skip.py:54
        s[P + "1.0.1." + c + ".bias"] = (cfg.skip[i],)
        _bn_shapes(P + "1.0.2", cfg.skip[i], s)                                # :55
        s[P + "1.1.1." + c + ".weight"] = (cfg.down[i], depth_in, 3, 3)        # :60 (stride 2)
        s[P + "1.1.1." + c + ".bias"] = (cfg.down[i],)
        _bn_shapes(P + "1.1.2", cfg.down[i], s)
        s[P + "1.1.4." + c + ".weight"] = (cfg.down[i], cfg.down[i], 3, 3)     # :64
        s[P + "1.1.4." + c + ".bias"] = (cfg.down[i],)
        _bn_shapes(P + "1.1.5", cfg.down[i], s)
        if not last:
            level(P + "1.1.7.", i + 1, cfg.down[i])                            # :74 deeper_main
            k = cfg.up[i + 1]
        else:
            k = cfg.down[i]
        _bn_shapes(P + "2", cfg.skip[i] + k, s)                                # :51
        s[P + "3." + c + ".weight"] = (cfg.up[i], cfg.skip[i] + k, 3, 3)       # :79
        s[P + "3." + c + ".bias"] = (cfg.up[i],)
        _bn_shapes(P + "4", cfg.up[i], s)
        s[P + "6." + c + ".weight"] = (cfg.up[i], cfg.up[i], 1, 1)             # :85
        s[P + "6." + c + ".bias"] = (cfg.up[i],)
        _bn_shapes(P + "7", cfg.up[i], s)

    level("", 0, cfg.input_depth)
    s["9." + c + ".weight"] = (cfg.n_channels, cfg.up[0], 1, 1)                # :92
    s["9." + c + ".bias"] = (cfg.n_channels,)
    # order keys the way nn.Module.state_dict() walks the tree (depth first, registration order)
    return s


def _conv(sd, key, x, k, stride, cfg, downsample_mode="stride"):
    """models/DIP/utils.py:83-105."""
    pool = None
    if stride != 1 and downsample_mode != "stride":                   # :86-94
        pool = F.avg_pool2d if downsample_mode == "avg" else F.max_pool2d
        pool_k, stride = stride, 1
    to_pad = int((k - 1) / 2)
    if cfg.pad == "reflection":
        if to_pad:
            x = F.pad(x, (to_pad,) * 4, mode="reflect")
        to_pad = 0
    y = F.conv2d(x, sd[key + ".weight"], sd[key + ".bias"], stride=stride, padding=to_pad)
    return pool(y, pool_k, pool_k) if pool is not None else y


def _act_of(cfg):
    """models/DIP/utils.py:62-76."""
    if cfg.act_fun == "LeakyReLU":
        return lambda x: F.leaky_relu(x, 0.2)
    if cfg.act_fun == "ELU":
        return F.elu
    return lambda x: x


def concat_center_crop(inputs):
    """models/DIP/utils.py:24-38."""
    h = min(t.shape[2] for t in inputs)
    w = min(t.shape[3] for t in inputs)
    out = []
    for t in inputs:
        d2 = (t.shape[2] - h) // 2
        d3 = (t.shape[3] - w) // 2
        out.append(t[:, :, d2:d2 + h, d3:d3 + w])
    return torch.cat(out, dim=1)


def skip_forward(sd, x, cfg, train=True):
    n = len(cfg.down)
    c = str(cfg.ci)
    _act = _act_of(cfg)

    def level(P, i, x):
        last = i == n - 1
        s = _conv(sd, P + "1.0.1." + c, x, 1, 1, cfg)
        s = _act(batch_norm(sd, P + "1.0.2", s, train))
        d = _conv(sd, P + "1.1.1." + c, x, 3, 2, cfg, cfg.downsample_mode)
        d = _act(batch_norm(sd, P + "1.1.2", d, train))
        d = _conv(sd, P + "1.1.4." + c, d, 3, 1, cfg)
        d = _act(batch_norm(sd, P + "1.1.5", d, train))
        if not last:
            d = level(P + "1.1.7.", i + 1, d)
        if cfg.upsample_mode == "bilinear":
            d = F.interpolate(d, scale_factor=2, mode="bilinear", align_corners=False)   # nn.Upsample default
        else:
            d = F.interpolate(d, scale_factor=2, mode="nearest")
        z = concat_center_crop([s, d])
        z = batch_norm(sd, P + "2", z, train)
        z = _conv(sd, P + "3." + c, z, 3, 1, cfg)
        z = _act(batch_norm(sd, P + "4", z, train))
        z = _conv(sd, P + "6." + c, z, 1, 1, cfg)
        z = _act(batch_norm(sd, P + "7", z, train))
        return z

    z = level("", 0, x)
    z = _conv(sd, "9." + c, z, 1, 1, cfg)
    return torch.sigmoid(z)                                                          # skip.py:93-94


def get_noise(input_depth, spatial_size, var=1.0 / 10):
    """utils/DIP.py:79-96 with method='noise', noise_type='u': U(0,1)*var from torch's
    global CPU generator (callers seed it)."""
    if isinstance(spatial_size, int):
        spatial_size = (spatial_size, spatial_size)
    t = torch.zeros([1, input_depth, spatial_size[0], spatial_size[1]])
    t.uniform_()
    t *= var
    return t
