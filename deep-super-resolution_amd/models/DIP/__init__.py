"""get_net with the reference's signature (/root/reference/models/DIP/__init__.py:8-18)."""
from .skip import skip


def get_net(input_depth, NET_TYPE, pad, upsample_mode, n_channels=3, act_fun='LeakyReLU', skip_n33d=128, skip_n33u=128,
            skip_n11=4, num_scales=5, downsample_mode='stride'):
    if NET_TYPE != 'skip':
        assert False
    as_list = lambda v: [v] * num_scales if isinstance(v, int) else v   # noqa: E731
    return skip(input_depth, n_channels, num_channels_down=as_list(skip_n33d), num_channels_up=as_list(skip_n33u),
                num_channels_skip=as_list(skip_n11), upsample_mode=upsample_mode, downsample_mode=downsample_mode,
                need_sigmoid=True, need_bias=True, pad=pad, act_fun=act_fun)
