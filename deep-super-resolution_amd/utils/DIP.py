"""DIP optimisation helpers with the reference's surface (/root/reference/utils/DIP.py)."""
import torch

from ..optim import FusedAdam


def optimize(optimizer_type, parameters, closure, learning_rate, num_iter):
    """utils/DIP.py:7-42.  'adam' runs the fused HIP Adam; the LBFGS branch is never selected by the reference's
    own caller (DIP.py:99 passes 'adam') and is out of scope."""
    if optimizer_type == 'adam':
        optimizer = FusedAdam(parameters, lr=learning_rate)
        for _ in range(num_iter):
            optimizer.zero_grad()
            closure()
            optimizer.step()
        optimizer.zero_grad(set_to_none=True)
        del optimizer
    elif optimizer_type == 'LBFGS':
        raise NotImplementedError("LBFGS is outside the hot path (never selected by DIP.py)")
    else:
        assert False


def get_params(opt_over, net, net_input, downsampler=None):
    """utils/DIP.py:44-68."""
    params = []
    for opt in opt_over.split(','):
        if opt == 'net':
            params += [x for x in net.parameters()]
        elif opt == 'down':
            assert downsampler is not None
            params = [x for x in downsampler.parameters()]
        elif opt == 'input':
            net_input.requires_grad = True
            params += [net_input]
        else:
            assert False, 'what is it?'
    return params


def fill_noise(x, noise_type):
    if noise_type == 'u':
        x.uniform_()
    elif noise_type == 'n':
        x.normal_()
    else:
        assert False


def get_noise(input_depth, method, spatial_size, noise_type='u', var=1. / 10):
    """utils/DIP.py:79-96 -- host-side, drawn from torch's CPU generator exactly like the reference."""
    if isinstance(spatial_size, int):
        spatial_size = (spatial_size, spatial_size)
    if method == 'noise':
        net_input = torch.zeros([1, input_depth, spatial_size[0], spatial_size[1]])
        fill_noise(net_input, noise_type)
        net_input *= var
    elif method == 'meshgrid':
        assert input_depth == 2
        ys = torch.arange(0, spatial_size[0], dtype=torch.float64) / float(spatial_size[0] - 1)
        xs = torch.arange(0, spatial_size[1], dtype=torch.float64) / float(spatial_size[1] - 1)
        net_input = torch.stack([xs[None, :].expand(spatial_size[0], -1), ys[:, None].expand(-1, spatial_size[1])])[None]
    else:
        assert False
    return net_input
