"""Oracle (test infrastructure): fp32 CPU restatement of the SRGAN generator and
discriminator as pure functions over a reference-format ``state_dict``.

Follows /root/reference:
  models/GAN/generator.py:4-25   ResidualBlock
  models/GAN/generator.py:27-41  PixelShuffleBlock
  models/GAN/generator.py:44-81  Generator
  models/GAN/discriminator.py:4-19, 21-74  DiscriminatorConvBlock / Discriminator
State is a plain dict name->tensor with the reference's key names; BatchNorm buffers
are updated in place in train mode exactly like nn.BatchNorm2d (momentum 0.1, eps 1e-5,
unbiased variance for the running estimate).
"""
import math

import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


# ----------------------------------------------------------------------------- shapes
def _bn_shapes(prefix, c, out):
    out[prefix + ".weight"] = (c,)
    out[prefix + ".bias"] = (c,)
    out[prefix + ".running_mean"] = (c,)
    out[prefix + ".running_var"] = (c,)
    out[prefix + ".num_batches_tracked"] = ()


def generator_shapes(factor=8, residual_blocks_count=16):
    """Key -> shape of Generator(factor).state_dict() (generator.py:45-64).

    factor = 2**k gives k pixel-shuffle blocks; the reference itself only accepts
    8 (k=3) and 16 (k=4) (generator.py:55-58)."""
    k = int(round(math.log2(factor)))
    assert 2 ** k == factor and k >= 1
    s = {}
    s["conv1.weight"] = (64, 3, 9, 9)
    s["conv1.bias"] = (64,)
    s["prelu1.weight"] = (1,)
    for i in range(residual_blocks_count):
        p = f"residual_blocks.{i}"
        s[p + ".conv1.weight"] = (64, 64, 3, 3)
        s[p + ".conv1.bias"] = (64,)
        _bn_shapes(p + ".bn1", 64, s)
        s[p + ".prelu1.weight"] = (1,)
        s[p + ".conv2.weight"] = (64, 64, 3, 3)
        s[p + ".conv2.bias"] = (64,)
        _bn_shapes(p + ".bn2", 64, s)
    s["conv2.weight"] = (64, 64, 3, 3)
    s["conv2.bias"] = (64,)
    _bn_shapes("bn1", 64, s)
    for j in range(k):
        p = f"pixel_shuffle_blocks.{j}"
        s[p + ".conv1.weight"] = (256, 64, 3, 3)
        s[p + ".conv1.bias"] = (256,)
        s[p + ".prelu1.weight"] = (1,)
    s["conv3.weight"] = (3, 64, 9, 9)
    s["conv3.bias"] = (3,)
    return s


D_BLOCKS = [(64, 64, 2), (64, 128, 1), (128, 128, 2), (128, 256, 1),
            (256, 256, 2), (256, 512, 1), (512, 512, 2)]   # discriminator.py:29-35


def discriminator_flat_features(hr_shape):
    """discriminator.py:48-56 -- 512 * H_out * W_out after four stride-2 3x3 pad-1 convs."""
    h, w = hr_shape
    for _ in range(4):
        h = (h + 2 - 3) // 2 + 1
        w = (w + 2 - 3) // 2 + 1
    return 512 * h * w


def discriminator_shapes(hr_shape):
    s = {}
    s["conv.weight"] = (64, 3, 3, 3)
    s["conv.bias"] = (64,)
    for i, (ci, co, _) in enumerate(D_BLOCKS):
        p = f"convblocks.{i}"
        s[p + ".conv1.weight"] = (co, ci, 3, 3)
        s[p + ".conv1.bias"] = (co,)
        _bn_shapes(p + ".bn1", co, s)
    flat = discriminator_flat_features(hr_shape)
    s["dense1.weight"] = (1024, flat)
    s["dense1.bias"] = (1024,)
    s["dense2.weight"] = (1, 1024)
    s["dense2.bias"] = (1,)
    return s


def template(shapes):
    """Zero state_dict with the right dtypes (num_batches_tracked is int64)."""
    return {k: torch.zeros(v, dtype=torch.int64 if k.endswith("num_batches_tracked") else torch.float32)
            for k, v in shapes.items()}


# ------------------------------------------------------------------------------- ops
def batch_norm(sd, prefix, x, train):
    """nn.BatchNorm2d forward with default momentum/eps, buffers updated in place."""
    w, b = sd[prefix + ".weight"], sd[prefix + ".bias"]
    rm, rv = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    if train:
        mean = x.mean(dim=(0, 2, 3))
        var = x.var(dim=(0, 2, 3), unbiased=False)
        n = x.numel() // x.shape[1]
        with torch.no_grad():
            rm.mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * mean.detach())
            rv.mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * var.detach() * (n / max(n - 1, 1)))
            sd[prefix + ".num_batches_tracked"] += 1
    else:
        mean, var = rm, rv
    xh = (x - mean[None, :, None, None]) * torch.rsqrt(var[None, :, None, None] + BN_EPS)
    return xh * w[None, :, None, None] + b[None, :, None, None]


def prelu(x, a):
    return torch.where(x >= 0, x, a * x)   # nn.PReLU(num_parameters=1)


def pixel_shuffle2(x):
    """out[n,c,2h+i,2w+j] = in[n,4c+2i+j,h,w] (nn.PixelShuffle(2))."""
    n, c4, h, w = x.shape
    c = c4 // 4
    return x.view(n, c, 2, 2, h, w).permute(0, 1, 4, 2, 5, 3).reshape(n, c, 2 * h, 2 * w)


def residual_block(sd, p, x, train):
    z = F.conv2d(x, sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], padding=1)   # generator.py:15
    z = batch_norm(sd, p + ".bn1", z, train)                                     # :17
    z = prelu(z, sd[p + ".prelu1.weight"])                                       # :18
    z = F.conv2d(z, sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], padding=1)   # :20
    z = batch_norm(sd, p + ".bn2", z, train)                                     # :21
    return x + z                                                                 # :23


def generator_forward(sd, x, train=True):
    """generator.py:66-81.  The number of residual / pixel-shuffle blocks is read off the keys."""
    nres = len({k.split(".")[1] for k in sd if k.startswith("residual_blocks.")})
    nps = len({k.split(".")[1] for k in sd if k.startswith("pixel_shuffle_blocks.")})
    z = F.conv2d(x, sd["conv1.weight"], sd["conv1.bias"], padding=4)
    x0 = prelu(z, sd["prelu1.weight"])
    z = x0
    for i in range(nres):
        z = residual_block(sd, f"residual_blocks.{i}", z, train)
    z = F.conv2d(z, sd["conv2.weight"], sd["conv2.bias"], padding=1)
    z = batch_norm(sd, "bn1", z, train)
    z = x0 + z
    for j in range(nps):
        p = f"pixel_shuffle_blocks.{j}"
        z = F.conv2d(z, sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], padding=1)  # :37
        z = pixel_shuffle2(z)                                                       # :38
        z = prelu(z, sd[p + ".prelu1.weight"])                                      # :39
    z = F.conv2d(z, sd["conv3.weight"], sd["conv3.bias"], padding=4)
    return torch.tanh(z)


def discriminator_forward(sd, x, train=True):
    """discriminator.py:58-74 (flatten is C,H,W order: x.view(N,-1) on NCHW, :65)."""
    x = F.conv2d(x, sd["conv.weight"], sd["conv.bias"], padding=1)
    x = F.leaky_relu(x, 0.2)
    for i, (_, _, stride) in enumerate(D_BLOCKS):
        p = f"convblocks.{i}"
        x = F.conv2d(x, sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], stride=stride, padding=1)
        x = batch_norm(sd, p + ".bn1", x, train)
        x = F.leaky_relu(x, 0.2)
    x = x.reshape(x.shape[0], -1)
    x = F.linear(x, sd["dense1.weight"], sd["dense1.bias"])
    x = F.leaky_relu(x, 0.2)
    x = F.linear(x, sd["dense2.weight"], sd["dense2.bias"])
    return torch.sigmoid(x)


def trainable(sd):
    """Keys that nn.Module.parameters() would yield, in registration order."""
    return [k for k in sd if not (k.endswith("running_mean") or k.endswith("running_var")
                                  or k.endswith("num_batches_tracked"))]
