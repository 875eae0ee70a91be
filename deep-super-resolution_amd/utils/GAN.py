"""GAN losses with the reference's surface (/root/reference/utils/GAN.py), on the HIP path.

  Vgg19Loss (:7-92)          VGG19 ``features[:36]`` trunk (frozen) + MSE of the two feature maps, both images
                             preprocessed by torchvision's ImageClassification preset (:82-83)
  get_adversarial_loss (:96) BCE(fake, 1)
  get_loss_D (:101-105)      BCE(real, 1) + BCE(fake, 0)
  PerceptualLoss (:108-124)  content + adversarial, unweighted

torchvision is not available here and its IMAGENET1K_V1 weights cannot be downloaded, so:
  * the trunk architecture is stated explicitly (same layer indices => same state_dict keys ``net.0.<i>.*``);
    weights are deterministic stand-ins unless ``Vgg19Loss(state_dict=...)`` is given a torchvision
    ``vgg19().features`` state_dict loaded by the caller from a local file;
  * transforms() is restated from the published preset: resize shorter side to 256 (bilinear, antialias),
    centre-crop 224, normalise.  The antialias weight tables follow ATen's upsample_bilinear2d_aa.
"""
import ctypes as C

import numpy as np
import torch
import torch.nn as nn

from .. import functional as F

VGG19_CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M', 512, 512, 512, 512]
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def aa_weights(in_size, out_size):
    """ATen's antialiased bilinear (triangle filter) source windows: list of (start, weights float32)."""
    scale = in_size / out_size
    support = scale if scale >= 1.0 else 1.0            # (interp_size/2) * max(scale, 1), interp_size = 2
    invscale = 1.0 / scale if scale >= 1.0 else 1.0
    out = []
    for i in range(out_size):
        center = scale * (i + 0.5)
        xmin = max(int(center - support + 0.5), 0)
        xsize = min(int(center + support + 0.5), in_size) - xmin
        j = np.arange(xsize, dtype=np.float64)
        w = np.clip(1.0 - np.abs((j + xmin - center + 0.5) * invscale), 0.0, None)
        tot = w.sum()
        w = w / tot if tot != 0 else w
        out.append((xmin, w.astype(np.float32)))
    return out


def _pack_tables(windows, kt):
    n = len(windows)
    s = np.zeros(n, dtype=np.int32)
    c = np.zeros(n, dtype=np.int32)
    w = np.zeros((n, kt), dtype=np.float32)
    for i, (st, ws) in enumerate(windows):
        s[i], c[i] = st, len(ws)
        w[i, :len(ws)] = ws
    return s, c, w


def _transpose_windows(windows, in_size):
    """For every input index: the contiguous run of outputs that read it, with their weights."""
    lists = [[] for _ in range(in_size)]
    for o, (st, ws) in enumerate(windows):
        for j, wv in enumerate(ws):
            lists[st + j].append((o, wv))
    out = []
    for lst in lists:
        if not lst:
            out.append((0, np.zeros(0, dtype=np.float32)))
            continue
        o0, o1 = lst[0][0], lst[-1][0]
        w = np.zeros(o1 - o0 + 1, dtype=np.float32)
        for o, wv in lst:
            w[o - o0] += wv
        out.append((o0, w))
    return out


class ResampleTables:
    """Device-resident tables of the separable resize(->resize_to)+centre-crop(crop) for one input size."""

    def __init__(self, in_h, in_w, device, resize_to=256, crop=224, mean=IMAGENET_MEAN, std=IMAGENET_STD):
        if in_h <= in_w:
            nh, nw = resize_to, int(resize_to * in_w / in_h)
        else:
            nh, nw = int(resize_to * in_h / in_w), resize_to
        top, left = int(round((nh - crop) / 2.0)), int(round((nw - crop) / 2.0))
        wy = aa_weights(in_h, nh)[top:top + crop]
        wx = aa_weights(in_w, nw)[left:left + crop]
        ty, tx = _transpose_windows(wy, in_h), _transpose_windows(wx, in_w)
        self.kt = max(max(len(w) for _, w in t) for t in (wy, wx, ty, tx))
        self.in_h, self.in_w, self.out_h, self.out_w = in_h, in_w, len(wy), len(wx)

        def dev(arrs):
            return [torch.from_numpy(a).to(device) for a in arrs]

        self.ys, self.yc, self.yw = dev(_pack_tables(wy, self.kt))
        self.xs, self.xc, self.xw = dev(_pack_tables(wx, self.kt))
        self.tys, self.tyc, self.tyw = dev(_pack_tables(ty, self.kt))
        self.txs, self.txc, self.txw = dev(_pack_tables(tx, self.kt))
        self.mean_c = (C.c_float * 3)(*mean)
        self.std_c = (C.c_float * 3)(*std)
        self.host = (wy, wx)


def _standin_vgg_state(seed=1234):
    """Deterministic frozen weights for the stand-in trunk: He-uniform so activations keep O(1) scale."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    cin = 3
    idx = 0
    for v in VGG19_CFG:
        if v == 'M':
            idx += 1
            continue
        bound = float(np.sqrt(6.0 / (cin * 9)))
        sd[f"{idx}.weight"] = (torch.rand(v, cin, 3, 3, generator=g) * 2 - 1) * bound
        sd[f"{idx}.bias"] = (torch.rand(v, generator=g) * 2 - 1) * 0.05
        cin = v
        idx += 2
    return sd


class Vgg19Loss(nn.Module):
    def __init__(self, state_dict=None, resize_to=256, crop=224):
        super(Vgg19Loss, self).__init__()
        layers = []
        cin = 3
        for v in VGG19_CFG:                       # torchvision vgg19().features[:36] (utils/GAN.py:19-57, :72)
            if v == 'M':
                layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
            else:
                layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
                cin = v
        assert len(layers) == 36
        self.net = nn.Sequential(nn.Sequential(*layers))          # same nesting as the reference => keys net.0.<i>.*
        self.net[0].load_state_dict(state_dict if state_dict is not None else _standin_vgg_state())
        self.pretrained = state_dict is not None
        self.mse = nn.MSELoss()
        for param in self.net.parameters():                       # :77-78
            param.requires_grad = False
        self.resize_to, self.crop = resize_to, crop
        self.compute_dtype = torch.bfloat16
        self._tables = {}

    def tables(self, h, w, device):
        key = (h, w, str(device))
        if key not in self._tables:
            self._tables[key] = ResampleTables(h, w, device, self.resize_to, self.crop)
        return self._tables[key]

    def features(self, image):
        x = F.ResizeNorm.apply(image, self.tables(image.shape[2], image.shape[3], image.device), self.compute_dtype)
        mods = list(self.net[0].children())
        i = 0
        # The trunk is a chain: every ReLU output has exactly one consumer (the next conv or a max-pool), whose backward
        # launch applies that ReLU's mask on its way out (functional.ActLink) -- 15 of the 16 activation-backward passes of
        # the trunk disappear; the last layer's output feeds the loss and keeps its own.
        link = None                   # the activation that produced the current x
        while i < len(mods):
            m = mods[i]
            if isinstance(m, nn.Conv2d):
                out_link = F.ActLink(F.ACT_RELU) if (torch.is_grad_enabled() and F.ACT_LINKS) else None
                x = F.ConvAct.apply(x, m.weight, m.bias, None,
                                    dict(stride=1, pad=1, act=F.ACT_RELU, in_link=link, out_link=out_link))   # conv + ReLU fused
                link = out_link
                i += 2
            else:
                x = F.MaxPool2.apply(x, link)
                link = None
                i += 1
        return F.ToNCHW.apply(x, 512)

    def target_features(self, image2):
        """Features of the HR target (:84-88): no gradient is ever used, so they can be computed ahead of time."""
        with torch.no_grad():
            return self.features(image2)

    def forward(self, image1, image2, features2=None):
        feature_map1 = self.features(image1)                      # :82,86
        if features2 is not None:
            feature_map2 = features2
        elif image2.requires_grad:
            feature_map2 = self.features(image2)
        else:
            feature_map2 = self.target_features(image2)
        return F.mse_loss(feature_map1, feature_map2)             # :90


def get_adversarial_loss(fake_output, bce_loss=None):
    return F.bce_const(fake_output, 1.0)


def get_loss_D(real_output, fake_output, bce_loss=None):
    return F.add_losses(F.bce_const(real_output, 1.0), F.bce_const(fake_output, 0.0))


class PerceptualLoss(nn.Module):
    def __init__(self, vgg_state_dict=None, resize_to=256, crop=224):
        super(PerceptualLoss, self).__init__()
        self.vgg_loss = Vgg19Loss(vgg_state_dict, resize_to, crop)

    def content(self, fake_output_G, HR_images, hr_features=None):
        return self.vgg_loss(fake_output_G, HR_images, hr_features)   # :119

    def adversarial(self, fake_output_D, bce_loss=None):
        return get_adversarial_loss(fake_output_D, bce_loss)      # :120

    def forward(self, fake_output_G, HR_images, fake_output_D, bce_loss=None):
        content_loss = self.content(fake_output_G, HR_images)
        adversarial_loss_ = self.adversarial(fake_output_D, bce_loss)
        return F.add_losses(content_loss, adversarial_loss_)      # unweighted sum (:122)
