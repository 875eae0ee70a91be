"""Deep-Image-Prior optimisation helpers behind the reference's surface (/root/reference/utils/DIP.py:7-105):
``optimize``, ``get_params``, ``fill_noise``, ``get_noise`` -- same names, argument meaning and error behaviour
(bare ``assert`` on an unknown option), written around this package's fused HIP Adam.

Host-side glue: the closure the caller passes in runs the HIP network; nothing here touches activations."""
import numpy as np
import torch

from ..optim import FusedAdam

# utils/DIP.py:21 -- the LBFGS branch first takes 100 Adam steps at this fixed rate
LBFGS_WARMUP_STEPS = 100
LBFGS_WARMUP_LR = 0.001


def _adam_loop(parameters, closure, learning_rate, steps):
    """zero_grad(); closure(); step()  `steps` times with the fused multi-tensor Adam kernel (utils/DIP.py:33-38)."""
    optimizer = FusedAdam(parameters, lr=learning_rate)
    for _ in range(steps):
        optimizer.zero_grad()
        closure()
        optimizer.step()
    optimizer.zero_grad(set_to_none=True)


def _lbfgs(parameters, closure, learning_rate, num_iter):
    """utils/DIP.py:19-31: Adam warm-up, then ONE torch.optim.LBFGS.step of ``max_iter=num_iter`` inner iterations
    with both tolerances disabled.  The closure (forward, loss, backward) is the HIP path; the two-loop recursion
    itself is torch's own vector arithmetic on the flattened parameters, exactly as in the reference.  LBFGS writes
    the parameters in place through torch ops, which bumps their version counters, so the packed 16-bit weight
    images are refreshed on the next forward."""
    parameters = list(parameters)
    _adam_loop(parameters, closure, LBFGS_WARMUP_LR, LBFGS_WARMUP_STEPS)
    optimizer = torch.optim.LBFGS(parameters, max_iter=num_iter, lr=learning_rate, tolerance_grad=-1,
                                  tolerance_change=-1)

    def closure2():
        optimizer.zero_grad()
        return closure()

    optimizer.step(closure2)


def optimize(optimizer_type, parameters, closure, learning_rate, num_iter):
    """Run the optimisation loop: ``'adam'`` (what DIP.py:99 selects) or ``'LBFGS'``; anything else asserts."""
    runners = {'adam': lambda: _adam_loop(list(parameters), closure, learning_rate, num_iter),
               'LBFGS': lambda: _lbfgs(parameters, closure, learning_rate, num_iter)}
    assert optimizer_type in runners
    runners[optimizer_type]()


def get_params(opt_over, net, net_input, downsampler=None):
    """Tensors to optimise over for a comma-separated ``opt_over`` of 'net', 'down', 'input' (utils/DIP.py:44-68).

    Bug-compatible with the reference on one point: 'down' REPLACES whatever was collected before it instead of
    extending it (utils/DIP.py:61 assigns), so "net,down" yields the downsampler's parameters only."""
    params = []
    for what in opt_over.split(','):
        if what == 'net':
            params = params + list(net.parameters())
        elif what == 'down':
            assert downsampler is not None
            params = list(downsampler.parameters())
        elif what == 'input':
            net_input.requires_grad = True
            params = params + [net_input]
        else:
            assert False, 'what is it?'
    return params


_FILLERS = {'u': torch.Tensor.uniform_, 'n': torch.Tensor.normal_}


def fill_noise(x, noise_type):
    """In-place U(0,1) ('u') or N(0,1) ('n') from torch's generator of x's device (utils/DIP.py:70-77)."""
    assert noise_type in _FILLERS
    _FILLERS[noise_type](x)


def get_noise(input_depth, method, spatial_size, noise_type='u', var=1. / 10):
    """[1, input_depth, H, W] network input (utils/DIP.py:79-105): 'noise' = noise * var drawn on the CPU default
    generator like the reference (DIP.py:32 then moves it to the device), 'meshgrid' = the two normalised coordinate
    planes (float64, x first), which needs input_depth == 2."""
    h, w = (spatial_size, spatial_size) if isinstance(spatial_size, int) else spatial_size[:2]
    if method == 'noise':
        net_input = torch.zeros([1, input_depth, h, w])
        fill_noise(net_input, noise_type)
        return net_input.mul_(var)
    if method == 'meshgrid':
        assert input_depth == 2
        xs, ys = np.meshgrid(np.arange(0, w) / float(w - 1), np.arange(0, h) / float(h - 1))
        return torch.from_numpy(np.stack([xs, ys]))[None]
    assert False
