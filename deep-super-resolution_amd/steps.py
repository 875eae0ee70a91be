"""The timed step recipes on the HIP path.

gen_l1_step : BASELINE config 2 (generator-only x4, L1, Adam) -- defined by SURVEY.md 8(d).
gan_step    : train_GAN.py:38-71 (do_epoch): D step, then G step with the detached adversarial term.
dip_step    : DIP.py:47-95 closure + utils/DIP.py:33-40 Adam iteration.
"""
import os

import torch

from . import functional as F


def gen_l1_step(gen, opt, lr_patches, hr_patches):
    fake = gen(lr_patches)
    loss = F.l1_loss(fake, hr_patches)
    opt.zero_grad()
    with F.batched_wgrad():              # the 35 3x3 weight gradients of the generator: one grouped launch at the end
        loss.backward()
    opt.step()
    return loss.detach(), fake.detach()


_side_streams = {}


def _side_stream(device, which="d"):
    key = (device.type, device.index, which)
    if key not in _side_streams:
        # DSR_SIDE_PRIORITY (tuning switch): HIP stream priority of the D-half stream, 0 = default, -1 = high
        _side_streams[key] = torch.cuda.Stream(device=device, priority=int(os.environ.get("DSR_SIDE_PRIORITY", "0")))
    return _side_streams[key]


TRACE = None    # development aid: a list -> gan_step appends (label, event recorded on the stream that reached that point)


def _mark(label, stream):
    if TRACE is not None:
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(stream)
        TRACE.append((label, ev))


def gan_step(gen, disc, perceptual, opt_g, opt_d, lr_patches, hr_patches, sync_g=None, sync_d=None, overlap=True,
             batch_wgrad=None):
    """train_GAN.py:38-71.  Returns (loss_D, loss_G, fake) as device tensors (no host sync here).

    Once `fake` exists the reference's two halves are independent: the D step (:44-53) reads only `fake.detach()`,
    and the G step's gradient comes only from the VGG content term -- the adversarial term is computed from a
    DETACHED generator output (:58), so it contributes a number to loss_G (:59) and nothing to loss_G.backward()
    (:63).  With `overlap` the D step (+ the no_grad D pass that feeds that number) runs on a second HIP stream
    while VGG forward/backward, the generator backward and its Adam run on the main one: the HBM-bound BatchNorm /
    Adam passes of one half execute under the MFMA-bound convolutions of the other.  The arithmetic is unchanged."""
    # train_GAN.py:46 and :56 evaluate gen(lr_patches) twice with the same weights and batch statistics -- the
    # two outputs are bit-identical and only the BatchNorm running statistics notice the second call.  One forward
    # (with the autograd graph the G step needs) + a double running-stat update is exactly equivalent.
    if batch_wgrad is None:
        batch_wgrad = os.environ.get("DSR_WGRAD_BATCH", "1") != "0"    # tuning switch: 0 = one weight-gradient launch per layer
    main = torch.cuda.current_stream(hr_patches.device)
    side = _side_stream(hr_patches.device) if overlap else main
    hr_feat = real_feat = None
    _mark("start", main)
    if overlap:
        # the VGG features of the HR target depend on nothing but the batch: they run beside the generator forward
        side.wait_stream(main)
        with torch.cuda.stream(side):
            hr_feat = perceptual.vgg_loss.target_features(hr_patches)
            hr_feat_ready = torch.cuda.Event()
            hr_feat_ready.record(side)
            # ... and so does D's conv trunk on the real batch (:44): same weights, same BatchNorm bookkeeping order
            # (real before fake) as in the reference, just started while the generator is still running
            real_feat = disc.features(hr_patches)
            _mark("side: HR VGG features + D(real) trunk done", side)
    gen.bn_updates = 2
    fake = gen(lr_patches)                                       # :46 and :56
    gen.bn_updates = 1
    _mark("main: G forward done", main)
    fake_det = fake.detach()
    if overlap:
        side.wait_stream(main)                                   # `fake` is complete before D reads it

    def d_half():
        real_d, fake_d = disc.forward_pair(hr_patches, fake_det, fa=real_feat)   # :44, :47 (BN statistics per batch, as there)
        loss_d = F.add_losses(F.bce_const(real_d, 1.0), F.bce_const(fake_d, 0.0))  # :48, utils/GAN.py:101-105
        opt_d.zero_grad()                                        # :51 (gan_D.zero_grad())
        with F.batched_wgrad(batch_wgrad):                       # (D's stride-1 layers, real + generated batch summed)
            loss_d.backward()                                    # :52
        _mark("D: backward done", torch.cuda.current_stream())
        if sync_d is not None:
            sync_d()
        opt_d.step()                                             # :53
        _mark("D: Adam done", torch.cuda.current_stream())
        with torch.no_grad():
            # :58 detaches the generator output, so this D pass never sends a gradient anywhere that survives
            # (D's .grad from it is wiped by the next zero_grad, :51); it still updates D's BN running statistics.
            adv = perceptual.adversarial(disc(fake_det))         # the adversarial number of :59
        return loss_d.detach(), adv

    if overlap:
        with torch.cuda.stream(side):
            loss_d, adv = d_half()
            _mark("D half done (incl. adversarial pass)", side)
    else:
        loss_d, adv = d_half()
    # --- generator (main stream)
    if hr_feat is not None:
        main.wait_event(hr_feat_ready)   # only the target features are needed here, not the D half queued behind them
        hr_feat.record_stream(main)
    content = perceptual.content(fake, hr_patches, hr_feat)      # the only term of :59 with a gradient path
    opt_g.zero_grad()                                            # :62
    _mark("main: VGG content loss forward done", main)
    # the generator's 35 3x3 layers: grouped launches.  With `overlap` every DSR_WGRAD_EARLY (default 16, 0 = off) collected
    # layers go out at once on a third stream: the generator half is the longer chain of the two, its backward is one long
    # dependency chain of input gradients, and the weight gradients hang off that chain -- launched at the exit they ran alone
    # at the end of the step (the D half has finished by then); launched as they become ready they run beside the chain
    early = int(os.environ.get("DSR_WGRAD_EARLY", "16")) if (overlap and batch_wgrad) else 0
    with F.batched_wgrad(batch_wgrad, early_stream=_side_stream(hr_patches.device, "wgrad") if early else None, early_every=early):
        content.backward()                                       # :63 (d adversarial / d generator == 0, see above)
    _mark("main: G backward done", main)
    if sync_g is not None:
        sync_g()
    opt_g.step()                                                 # :64
    _mark("main: G Adam done", main)
    if overlap:
        main.wait_stream(side)                                   # D half done before anything later on `main`
        loss_d.record_stream(main)
        adv.record_stream(main)
    loss_g = F.add_losses(content.detach(), adv)                  # :59 value (unweighted sum, utils/GAN.py:122)
    return loss_d, loss_g, fake_det


class GraphedStep:
    """A whole step (forward, backward, fused Adam) captured once in a HIP graph and replayed.

    The small workloads are launch-bound (config 2: ~500 kernel launches for 8 ms of work; a DIP iteration likewise), and
    everything a step does is capturable by construction: the C ABI only enqueues on the given stream, never
    allocates or synchronises, the Adam step counter and the BatchNorm batch counters live on the device, and
    scratch comes from torch's allocator (graph-private pool during capture).  `fn` must read its inputs from fixed
    tensors and return tensors; the returned tensors are the graph's static outputs (overwritten by each replay)."""

    def __init__(self, fn, warmup=3):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                    # warm-up off the default stream, as graph capture requires
            for _ in range(warmup):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = fn()

    def __call__(self):
        self.graph.replay()
        return self.out


class DipRunner:
    """DIP.py:22-123 state: fixed noise input, jitter buffer, Lanczos downsampler, Adam over the net."""

    def __init__(self, net, downsampler, net_input, lr_image, learning_rate, reg_noise_std, loss_scale=None):
        from .optim import FusedAdam
        # fp16 storage (the DIP default, see models/DIP/skip.py) needs a static loss scale so that activation
        # gradients (~1e-5 at the MSE) stay out of fp16's subnormal range; Adam un-scales on the fly.
        if loss_scale is None:
            loss_scale = 1024.0 if getattr(net, "compute_dtype", None) == torch.float16 else 1.0
        self.loss_scale = float(loss_scale)
        self.net, self.down = net, downsampler
        self.net_input_saved = net_input.detach().clone()        # DIP.py:33
        self.noise = net_input.detach().clone()                  # DIP.py:34
        self.net_input = net_input
        self.lr_image = lr_image
        self.sigma = reg_noise_std
        self.opt = FusedAdam(list(net.parameters()), lr=learning_rate,
                             grad_scale=1.0 / self.loss_scale)       # get_params('net') + utils/DIP.py:34

    def step(self, noise=None):
        """optimizer.zero_grad(); closure(); optimizer.step()  (utils/DIP.py:35-38, DIP.py:47-68)."""
        self.opt.zero_grad()
        if self.sigma > 0:
            if noise is None:
                noise = self.noise.normal_()                     # DIP.py:52 (torch RNG on whatever device holds it)
            self.net_input = self.net_input_saved + noise * self.sigma
        out_hr = self.net(self.net_input)                        # :60
        out_lr = self.down(out_hr)                               # :62
        loss = F.mse_loss(out_lr, self.lr_image)                 # :65
        with F.batched_wgrad():
            F.scale_loss(loss, self.loss_scale).backward()       # :68
        self.opt.step()
        return loss.detach(), out_hr.detach()
