"""Autograd functions over the C ABI (include/dsr_hip.h): the device work behind the
reference's nn.Module surface.

Internal activation format: torch tensor [N, H, W, Cp] (NHWC, Cp = channels rounded up to 8),
dtype bfloat16 (training) or float16 (inference).  Parameters stay fp32 in the reference's
layouts, so state_dicts interchange with the reference.

PyTorch is only plumbing here: it owns device memory, the stream and the autograd tape; every
FLOP is a HIP kernel from csrc/.  Nothing in this file can run without the extension.
"""
import ctypes as C
import os
import weakref

import torch

from . import _lib
from ._lib import (ACT_ELU, ACT_LEAKY, ACT_NONE, ACT_PRELU, ACT_RELU, ACT_SIGMOID, ACT_TANH, BF16, F16,  # noqa: F401
                   PAD_REFLECT, PAD_REPLICATE, PAD_ZERO, ConvDesc, Epilogue, check)

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def r8(c):
    return (c + 7) // 8 * 8


def _dt(t):
    if t.dtype == torch.bfloat16:
        return BF16
    if t.dtype == torch.float16:
        return F16
    raise TypeError(f"activation dtype must be bfloat16 or float16, got {t.dtype}")


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_gpu(t):
    if not t.is_cuda:
        raise RuntimeError("deep-super-resolution_amd: tensors must live on the MI355X (cuda device); "
                           "there is no CPU implementation of this path")


# ----------------------------------------------------------------------------- per-launch timing (bench.py roofline leg)
KERNEL_LOG = None          # set to a list to record (kind, desc-tuple, start_event, end_event) per conv launch


_OPS = {"fwd": 0, "dgrad": 1, "wgrad": 2}


def _timed(kind, desc, fn, ep=None, name=None):
    """Run one C-ABI conv call; when KERNEL_LOG is a list, bracket it with HIP events on the launch stream."""
    if KERNEL_LOG is None:
        return fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = fn()
    e1.record()
    if name is None:
        name = _lib.lib().dsr_conv_kernel_name(C.byref(desc), _OPS[kind], C.byref(ep) if ep is not None else None).decode()
    KERNEL_LOG.append((kind, (desc.N, desc.H, desc.W, desc.Cin, desc.Cout, desc.KH, desc.KW, desc.stride, desc.pad), e0, e1, name))
    return rc


# ----------------------------------------------------------------------------- weight images
_pack_cache = {}


def _version(p):
    return (p._version, getattr(p, "_dsr_version", 0), p.data_ptr())


def bump(p):
    """Called by the fused optimiser after it rewrote a parameter through its raw pointer."""
    p._dsr_version = getattr(p, "_dsr_version", 0) + 1


def clear_pack_cache():
    _pack_cache.clear()


def check_prelu_slopes(module):
    """Raise if a one-parameter nn.PReLU of `module` that feeds a fused conv+PReLU launch (generator.py:48,34: the head
    and the PixelShuffle blocks) holds a slope <= 0: those launches keep only the activation OUTPUT, from which the
    backward can tell the branch for a positive slope only (see ConvAct).  One host sync; call it between steps."""
    import torch.nn as nn
    bad = [(name, float(m.weight.detach().min())) for name, m in module.named_modules()
           if isinstance(m, nn.PReLU) and not float(m.weight.detach().min()) > 0.0]
    if bad:
        raise RuntimeError("PReLU slope(s) not positive: " + ", ".join(f"{n}.weight={v:g}" for n, v in bad) +
                           " -- conv+PReLU launches that store only the activation output cannot back-propagate "
                           "through them (their gradients are NaN by construction)")


def packed_weights(weight, desc, dtype):
    """16-bit [T][Cout_p][Cin_p] (forward) and [T][Cin_p][Cout_p] (dgrad) images of an OIHW fp32 weight."""
    key = (id(weight), dtype, tuple(weight.shape))
    ver = _version(weight)
    hit = _pack_cache.get(key)
    # ids (and allocator addresses) are recycled once a tensor dies: a hit must be THIS tensor object
    if hit is not None and hit[0] == ver and hit[3]() is weight:
        return hit[1], hit[2]
    lib = _lib.lib()
    nf = lib.dsr_conv_packed_elems(C.byref(desc), 0)
    nd = lib.dsr_conv_packed_elems(C.byref(desc), 1)
    wf = torch.empty(nf, dtype=dtype, device=weight.device)
    wd = torch.empty(nd, dtype=dtype, device=weight.device)
    w = weight.detach()
    if not w.is_contiguous():
        w = w.contiguous()
    check(lib.dsr_conv_pack_weight(C.byref(desc), _ptr(w), _ptr(wf), _ptr(wd), _stream()))
    _pack_cache[key] = (ver, wf, wd, weakref.ref(weight))
    if len(_pack_cache) > 4096:          # dead entries of short-lived tensors (tests); live models are far smaller
        for k in [k for k, v in _pack_cache.items() if v[3]() is None]:
            del _pack_cache[k]
    return wf, wd


def repack_cached(params):
    """Refresh, in one launch per 48 tensors, the packed images of every conv weight in `params` that has some (called by
    the fused optimiser right after it rewrote them; the images are rewritten in place, on the current stream)."""
    by_dtype = {}
    for p in params:
        if p.dim() != 4 or not p.is_contiguous():
            continue
        for dtype in (torch.bfloat16, torch.float16):
            key = (id(p), dtype, tuple(p.shape))
            hit = _pack_cache.get(key)
            if hit is not None and hit[3]() is p and hit[0] != _version(p):
                by_dtype.setdefault(dtype, []).append((key, p, hit))
    lib = _lib.lib()
    for dtype, items in by_dtype.items():
        k = len(items)
        arr = lambda vals: (C.c_void_p * k)(*vals)
        ints = lambda vals: (C.c_int * k)(*vals)
        check(lib.dsr_conv_pack_weight_multi(
            BF16 if dtype == torch.bfloat16 else F16, k, arr([p.data_ptr() for _, p, _ in items]),
            arr([h[1].data_ptr() for _, _, h in items]), arr([h[2].data_ptr() for _, _, h in items]),
            ints([p.shape[0] for _, p, _ in items]), ints([p.shape[1] for _, p, _ in items]),
            ints([p.shape[2] * p.shape[3] for _, p, _ in items]), _stream()))
        for key, p, hit in items:
            _pack_cache[key] = (_version(p), hit[1], hit[2], hit[3])


def make_desc(x, cout, kh, kw, stride, pad, pad_mode, cin):
    n, h, w, cp = x.shape
    assert cp == r8(cin), (cp, cin)
    return ConvDesc(_dt(x), n, h, w, cin, cout, kh, kw, stride, pad, pad_mode)


def first2_supported(img, w0, w1, stride1):
    """The image layer + the stride-2 64 -> 64 layer behind it as one forward kernel (dsr_conv_first2_fwd)?"""
    if not (img.is_cuda and img.dim() == 4 and img.shape[-1] == 8 and tuple(w0.shape[2:]) == (3, 3) and tuple(w1.shape) == (64, 64, 3, 3)
            and w0.shape[0] == 64 and w0.shape[1] <= 8 and stride1 == 2):
        return False
    n, h, w, _ = img.shape
    d0 = ConvDesc(_dt(img), n, h, w, w0.shape[1], 64, 3, 3, 1, 1, PAD_ZERO)
    d1 = ConvDesc(_dt(img), n, h, w, 64, 64, 3, 3, 2, 1, PAD_ZERO)
    return bool(_lib.lib().dsr_conv_first2_supported(C.byref(d0), C.byref(d1)))


def _out_hw(desc):
    oh, ow = C.c_int(), C.c_int()
    check(_lib.lib().dsr_conv_out_size(C.byref(desc), C.byref(oh), C.byref(ow)))
    return oh.value, ow.value


def _scr():
    return _lib.lib().dsr_pw_scratch_rows()


def _reduce_blocks(p):
    rpb = C.c_int()
    blocks = _lib.lib().dsr_pw_reduce_blocks(p, C.byref(rpb))
    return blocks, rpb.value


_BN_BWD_BLOCKS = int(os.environ.get("DSR_PW_BN_REDUCE_BLOCKS", "1280"))


def _bn_bwd_blocks(p, act):
    """Grid of dsr_pw_bn_act_bwd_reduce.  Without the PReLU slope gradient the kernel fits five waves per SIMD: 1280 blocks
    (five per CU, all resident at once) stream 10-20 % faster than the 1024 of the other row reductions
    (tools/microbench_pw.py); the PReLU form (four waves per SIMD) keeps the common grid."""
    if act == ACT_PRELU:
        return _reduce_blocks(p)
    rpb = max(64, -(-p // _BN_BWD_BLOCKS))
    return -(-p // rpb), rpb


_wgrad_batch = None      # the open batched_wgrad context (per process: backward passes are issued from one thread here)


class batched_wgrad:
    """``with batched_wgrad(): loss.backward()`` -- every 3x3 stride-1 weight gradient of that backward pass is formed by ONE
    grouped contraction launch and one reduction launch when the block exits (dsr_conv_wgrad_batched), instead of one
    contraction + two reduction launches per layer.  The SRGAN trunk is 33 such layers with a single 64x64 output tile
    each (generator.py:7-25,52): alone, a layer can only fill the chip by cutting its pixels into ~256 chunks and
    reducing 37 MB of partials; together ~30 chunks per layer do.  A weight that appears twice in the graph (the
    discriminator on the real and the generated batch, train_GAN.py:44-47) gets both contributions summed inside the
    reduction, which also replaces autograd's elementwise accumulation.

    Inside the block a layer's backward returns its (still unwritten) gradient tensor, so nothing may READ a weight's
    .grad before the block exits: hooks on those parameters, and accumulation into an existing .grad -- a weight whose
    .grad is already set is therefore computed at once, unbatched.  On exit each parameter's .grad is checked to be the
    tensor the launch wrote (autograd may have cloned it) and repaired if not."""

    MAX_PROBLEMS = 256           # kMaxBatchProblems of dsr_conv_wgrad_batched: larger groups are split in flush()

    def __init__(self, enabled=True, early_stream=None, early_every=0):
        """early_stream / early_every: every `early_every` collected problems are launched AT ONCE on `early_stream` (behind an
        event of the stream the backward pass runs on) instead of at the exit: the grouped launch of a long chain of layers
        (the generator's trunk) then runs beside the rest of that chain's input gradients instead of after them.  Only for
        weights used once inside the block; the exit waits for the early launches."""
        self.enabled = enabled
        self.items = {}          # id(weight) -> [weight, returned tensor, alias of it, [(desc, x, dy), ...], [extra addends]]
        self.unbatched = set()   # id(weight) of weights that already returned a REAL gradient inside this block
        self.early_stream = early_stream if early_every > 0 else None
        self.early_every = int(early_every)
        self.done = []           # entries whose launch is already on early_stream
        self.early_ids = set()
        self.reused = set()      # ... and which were used again afterwards

    def __enter__(self):
        global _wgrad_batch
        self.outer = _wgrad_batch
        if self.enabled and self.outer is None:
            _wgrad_batch = self
        return self

    def __exit__(self, et, ev, tb):
        global _wgrad_batch
        if _wgrad_batch is self:
            _wgrad_batch = None
            if et is None:
                self.flush()
                self._finish_early()
            else:
                if self.done:
                    torch.cuda.current_stream().wait_stream(self.early_stream)
                # backward raised: the launch never ran, so a .grad that autograd already pointed at one of the placeholder
                # tensors holds uninitialised memory -- drop it rather than leave garbage behind
                for weight, _, alias, _, _ in self.items.values():
                    g = weight.grad
                    if g is not None and g.data_ptr() == alias.data_ptr():
                        weight.grad = None
            self.items = {}
            self.unbatched = set()
            self.done = []
            self.early_ids = set()
            self.reused = set()
        return False

    def add(self, weight, desc, x, dy, wshape):
        """The gradient tensor to return for this use of `weight` (None for a second use: its contribution joins the first)."""
        ent = self.items.get(id(weight))
        if ent is not None:
            ent[3].append((desc, x, dy))
            return None
        dw = torch.empty(wshape, dtype=torch.float32, device=x.device)
        # (the alias shares dw's storage through a tensor object of its own: autograd takes over a gradient only while
        # nobody else holds the very tensor it was handed)
        self.items[id(weight)] = [weight, None, dw.detach(), [(desc, x, dy)], []]
        if self.early_stream is not None and sum(len(e[3]) for e in self.items.values()) >= self.early_every:
            self._launch_early()
        return dw

    def batchable(self, weight):
        """False for a weight whose gradient launch is already under way on the early stream: a second use of it inside the
        block is computed on its own, after that launch (the caller's stream is made to wait for it here)."""
        if id(weight) in self.early_ids:
            torch.cuda.current_stream().wait_stream(self.early_stream)
            self.reused.add(id(weight))      # autograd now sums a real tensor onto the finished placeholder: .grad is that sum
            return False
        return id(weight) not in self.unbatched

    def _launch_early(self):
        cur = torch.cuda.current_stream()
        ev = torch.cuda.Event()
        ev.record(cur)                       # every dy collected so far was enqueued on `cur` before this point
        self.early_stream.wait_event(ev)
        ents = list(self.items.values())
        with torch.cuda.stream(self.early_stream):
            self._launch(ents)
        for weight, _, alias, uses, _ in ents:
            alias.record_stream(self.early_stream)
            for _, x, dy in uses:            # (allocated on `cur`: not to be handed out again before the early launch has read them)
                x.record_stream(self.early_stream)
                dy.record_stream(self.early_stream)
            self.early_ids.add(id(weight))
        self.done.extend(ents)
        self.items = {}

    def _finish_early(self):
        if not self.done:
            return
        torch.cuda.current_stream().wait_stream(self.early_stream)
        self._settle([e for e in self.done if id(e[0]) not in self.reused])

    def add_unbatchable(self, weight, dw):
        """A use of `weight` that the grouped launch cannot take (other stride / padding ...), computed at once as `dw`.  If an
        earlier use of the same weight joined the batch, autograd already holds its PLACEHOLDER: summing a real tensor into it
        would add garbage, and the later launch would overwrite the sum.  The real contribution is therefore kept here and
        added to the launch's result in flush(); the caller returns None for this use.  Returns True when that happened."""
        ent = self.items.get(id(weight))
        if ent is None:
            self.unbatched.add(id(weight))       # later batchable uses of this weight must not join the batch either
            return False
        ent[4].append(dw)
        return True

    def flush(self):
        if not self.items:
            return
        ents = list(self.items.values())
        self._launch(ents)
        self._settle(ents)

    def _launch(self, ents):
        lib = _lib.lib()
        by_dtype = {}
        for ent in ents:
            by_dtype.setdefault(ent[3][0][0].dtype, []).append(ent)
        groups = []
        for group in by_dtype.values():      # at most MAX_PROBLEMS problems per launch; the uses of one weight stay together
            cur, ncur = [], 0
            for ent in group:
                if cur and ncur + len(ent[3]) > self.MAX_PROBLEMS:
                    groups.append(cur)
                    cur, ncur = [], 0
                cur.append(ent)
                ncur += len(ent[3])
            if cur:
                groups.append(cur)
        for group in groups:
            descs, xs, dys, dws = [], [], [], []
            for weight, _, alias, uses, _ in group:
                for desc, x, dy in uses:
                    descs.append(desc)
                    xs.append(x.data_ptr())
                    dys.append(dy.data_ptr())
                    dws.append(alias.data_ptr())
            n = len(descs)
            darr = (_lib.ConvDesc * n)(*descs)
            xa, ya, wa = (C.c_void_p * n)(*xs), (C.c_void_p * n)(*dys), (C.c_void_p * n)(*dws)
            wsz = lib.dsr_conv_wgrad_batched_workspace(n, darr, wa)
            ws = torch.empty(max(wsz, 16), dtype=torch.uint8, device=group[0][2].device)
            if KERNEL_LOG is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            check(lib.dsr_conv_wgrad_batched(n, darr, xa, ya, wa, _ptr(ws), wsz, _stream()))
            if KERNEL_LOG is not None:
                e1.record()
                KERNEL_LOG.append(("wgrad_batch", [(d.N, d.H, d.W, d.Cin, d.Cout, d.KH, d.KW, d.stride, d.pad) for d in descs],
                                   e0, e1, "conv_wgrad_dma_batch_kernel"))

    @staticmethod
    def _settle(ents):
        for weight, _, alias, uses, extra in ents:
            for e in extra:                               # contributions of uses the grouped launch could not take
                alias.add_(e)
            g = weight.grad
            if g is None:
                continue                                  # (nobody kept the gradient: e.g. torch.autograd.grad)
            if g.data_ptr() != alias.data_ptr():
                g.copy_(alias)                            # autograd stored a copy made before the launch: refresh it


FIRST_BWD_RECOMPUTE = os.environ.get("DSR_FIRST_BWD_RECOMPUTE", "1") != "0"   # 0: the first-layer backward reads the stored activation

ACT_LINKS = True      # development switch: False makes every layer run its own activation-backward pass (A/B of ActLink)


class ActLink:
    """Hand-over of an activation's backward between two neighbouring autograd nodes: the node that CONSUMES an activation
    output x (a conv's input-gradient launch, a max-pool backward) can multiply its result by act'(x) on the way out
    (dsr_conv_dgrad_masked, dsr_maxpool2_relu_bwd); it then sets `premasked`, and the node that PRODUCED x skips its own
    activation-backward pass.  One link per activation and forward call; only for an x with exactly one consumer (the VGG19
    trunk, utils/GAN.py:19-57)."""
    __slots__ = ("act", "slope", "premasked")

    def __init__(self, act, slope=0.0):
        self.act, self.slope, self.premasked = act, float(slope), False


def _conv_backward(desc, x, dy, wd, need_dx, need_dw, weight_shape, weight=None, addend=None, in_link=None, bn_link=None):
    """dx (+ `addend`: the gradient that reaches x along a skip path, summed in the dgrad epilogue where the kernel can; or
    `in_link`: the activation that produced x, whose backward mask is folded into the dgrad stores where the kernel can; or
    `bn_link`: x is the output of a BatchNorm + LeakyReLU layer that left its raw conv output and affine map there -- the
    dgrad launch then also forms that layer's BatchNorm-backward sums, dsr_conv_dgrad_bn, and leaves the partial rows in
    the link) and dw."""
    lib = _lib.lib()
    dx = dw = None
    if need_dx:
        dx = torch.empty_like(x)
        if (bn_link is not None and "y" in bn_link and addend is None and DGRAD_BN and bn_link["act"] in (ACT_NONE, ACT_LEAKY)
                and bn_link["y"].shape == x.shape and lib.dsr_conv_dgrad_bn_supported(C.byref(desc))):
            rows = lib.dsr_conv_dgrad_bn_rows(C.byref(desc))
            cp = x.shape[-1]
            part = torch.empty((rows + _scr()) * 3 * cp, dtype=torch.float32, device=x.device)
            check(_timed("dgrad", desc, lambda: lib.dsr_conv_dgrad_bn(
                C.byref(desc), _ptr(dy), _ptr(wd), _ptr(dx), _ptr(bn_link["y"]), _ptr(bn_link["scale"]), _ptr(bn_link["shift"]),
                bn_link["act"], float(bn_link["slope"]), _ptr(part), _stream()), name="conv_dgrad_s2_kernel<bn>"))
            bn_link["part"] = (part, rows, dx.data_ptr())
        elif in_link is not None and addend is None and lib.dsr_conv_dgrad_masked_supported(C.byref(desc)):
            check(_timed("dgrad", desc, lambda: lib.dsr_conv_dgrad_masked(C.byref(desc), _ptr(dy), _ptr(wd), _ptr(x), in_link.act,
                                                                            in_link.slope, _ptr(dx), _stream())))
            in_link.premasked = True
        elif addend is not None and lib.dsr_conv_dgrad_add_supported(C.byref(desc)):
            addend = addend.contiguous()
            check(_timed("dgrad", desc, lambda: lib.dsr_conv_dgrad_add(C.byref(desc), _ptr(dy), _ptr(wd), _ptr(addend), _ptr(dx),
                                                                         _stream()), name="conv_c64_kernel<3>"))
            addend = None
        else:
            wsz = lib.dsr_conv_dgrad_workspace(C.byref(desc))
            ws = torch.empty(max(wsz, 16), dtype=torch.uint8, device=x.device)
            check(_timed("dgrad", desc, lambda: lib.dsr_conv_dgrad(C.byref(desc), _ptr(dy), _ptr(wd), _ptr(dx), _ptr(ws), wsz,
                                                                     _stream())))
    if addend is not None:
        dx = addend if dx is None else dx + addend
    if (need_dw and _wgrad_batch is not None and weight is not None and weight.grad is None
            and _wgrad_batch.batchable(weight)                                # (an earlier use returned a real gradient / was launched early)
            and not getattr(weight, "_post_accumulate_grad_hooks", None)      # (a hook would read the gradient at once)
            and lib.dsr_conv_wgrad_batchable(C.byref(desc))):
        return dx, _wgrad_batch.add(weight, desc, x, dy, weight_shape)
    if need_dw:
        dw = torch.empty(weight_shape, dtype=torch.float32, device=x.device)
        wsz = lib.dsr_conv_wgrad_workspace(C.byref(desc))
        ws = torch.empty(wsz, dtype=torch.uint8, device=x.device)
        check(_timed("wgrad", desc, lambda: lib.dsr_conv_wgrad(C.byref(desc), _ptr(x), _ptr(dy), _ptr(dw), _ptr(ws), wsz,
                                                                 _stream())))
        if _wgrad_batch is not None and weight is not None and _wgrad_batch.add_unbatchable(weight, dw):
            dw = None            # joins the batched contribution of the same weight when the block exits
    return dx, dw


def _colsum(dy, c):
    """fp32 [c] column sums of an NHWC tensor (bias gradient)."""
    lib = _lib.lib()
    p = dy.numel() // dy.shape[-1]
    cp = dy.shape[-1]
    blocks, rpb = _reduce_blocks(p)
    part = torch.empty((blocks + _scr()) * cp, dtype=torch.float32, device=dy.device)
    check(lib.dsr_pw_colsum(_dt(dy), _ptr(dy), p, cp, blocks, rpb, _ptr(part), _stream()))
    out = torch.empty(c, dtype=torch.float32, device=dy.device)
    check(lib.dsr_pw_sum_rows(_ptr(part), blocks, cp, 0, c, 1.0, _ptr(out), 0, 1, _stream()))
    return out


# ----------------------------------------------------------------------------- layout edges
class ToNHWC(torch.autograd.Function):
    """fp32 NCHW [N,C,H,W] -> 16-bit NHWC [N,H,W,r8(C)] (pad channels are zero)."""

    @staticmethod
    def forward(ctx, x, dtype):
        _need_gpu(x)
        x = x.contiguous().float()
        n, c, h, w = x.shape
        out = torch.empty((n, h, w, r8(c)), dtype=dtype, device=x.device)
        check(_lib.lib().dsr_pw_nchw_to_nhwc(_dt(out), _ptr(x), _ptr(out), n, c, h, w, r8(c), _stream()))
        ctx.c = c
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        n, h, w, cp = g.shape
        out = torch.empty((n, ctx.c, h, w), dtype=torch.float32, device=g.device)
        check(_lib.lib().dsr_pw_nhwc_to_nchw(_dt(g), _ptr(g), _ptr(out), n, ctx.c, h, w, cp, _stream()))
        return out, None


class ToNCHW(torch.autograd.Function):
    """16-bit NHWC -> fp32 NCHW with the first `c` channels."""

    @staticmethod
    def forward(ctx, x, c):
        x = x.contiguous()
        n, h, w, cp = x.shape
        out = torch.empty((n, c, h, w), dtype=torch.float32, device=x.device)
        check(_lib.lib().dsr_pw_nhwc_to_nchw(_dt(x), _ptr(x), _ptr(out), n, c, h, w, cp, _stream()))
        ctx.dtype = x.dtype
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous().float()
        n, c, h, w = g.shape
        out = torch.empty((n, h, w, r8(c)), dtype=ctx.dtype, device=g.device)
        check(_lib.lib().dsr_pw_nchw_to_nhwc(_dt(out), _ptr(g), _ptr(out), n, c, h, w, r8(c), _stream()))
        return out, None


# ----------------------------------------------------------------------------- conv + bias + activation
class ConvAct(torch.autograd.Function):
    """y = act(conv(x, W) + b), optionally stored through PixelShuffle(2).

    generator.py:47-48 (9x9 + PReLU), :30-39 (3x3 64->256 + PixelShuffle + PReLU: a one-parameter
    PReLU commutes with the shuffle permutation), discriminator.py:25-27 (3x3 + LeakyReLU),
    utils/GAN.py:19-57 (VGG 3x3 + ReLU), plain convs (act none).
    Backward derives act' from the stored OUTPUT (the pre-activation is never written to HBM), which identifies the
    branch only for a POSITIVE slope.  A LeakyReLU slope is a host constant and is checked here; a learned PReLU slope
    lives on the device: if it has crossed zero the backward kernel turns every gradient of the launch into NaN (never
    a silently wrong number) and ``check_prelu_slopes(module)`` names the parameter.  ConvBNAct has no such
    restriction (it re-derives the sign from the saved conv output)."""

    @staticmethod
    def forward(ctx, x, weight, bias, prelu, cfg):
        _need_gpu(x)
        if cfg.get("act", ACT_NONE) == ACT_LEAKY and not float(cfg.get("slope", 0.0)) > 0.0:
            raise ValueError(f"ConvAct: LeakyReLU slope {cfg.get('slope', 0.0)} must be > 0 (the activation gradient is "
                             "derived from the stored output)")
        x = x.contiguous()
        cout, cin, kh, kw = weight.shape
        desc = make_desc(x, cout, kh, kw, cfg["stride"], cfg["pad"], cfg.get("pad_mode", PAD_ZERO), cin)
        oh, ow = _out_hw(desc)
        wf, wd = packed_weights(weight, desc, x.dtype)
        ps = bool(cfg.get("pixel_shuffle", False))
        n = x.shape[0]
        if ps:
            y = torch.empty((n, 2 * oh, 2 * ow, r8(cout // 4)), dtype=x.dtype, device=x.device)
        else:
            y = torch.empty((n, oh, ow, r8(cout)), dtype=x.dtype, device=x.device)
        act = cfg.get("act", ACT_NONE)
        ep = Epilogue(act, float(cfg.get("slope", 0.0)), _ptr(prelu), _ptr(bias), None, int(ps), None)
        if not cfg.get("defer", False):
            check(_timed("fwd", desc, lambda: _lib.lib().dsr_conv_fwd(C.byref(desc), _ptr(x), _ptr(wf), C.byref(ep), _ptr(y),
                                                                        _stream()), ep))
        # (defer: nothing is launched here -- the layer that consumes y computes this layer inside its own forward kernel and
        #  fills y only if a backward pass will need it: ConvBNAct with cfg["first2"], the discriminator's first two layers)
        ctx.desc, ctx.cfg, ctx.ps, ctx.act = desc, cfg, ps, act
        opl = cfg.get("out_ps_link")
        if opl is not None:
            # the 9x9 tail that consumes y may run this layer's activation backward inside its input-gradient launch
            opl.clear()
            if ps and act == ACT_PRELU and prelu is not None and cout == 256:
                opl["prelu"] = prelu.detach()
        ctx.wshape = tuple(weight.shape)
        ctx.weight_ref = weight
        ctx.weight_version = _version(weight)
        ctx.has_bias = bias is not None
        ctx.bias_ref = bias
        ctx.save_for_backward(x, y, wd, prelu if prelu is not None else torch.empty(0, device=x.device))
        return y

    @staticmethod
    def backward(ctx, dout):
        x, y, wd, prelu = ctx.saved_tensors
        prelu = prelu if prelu.numel() else None
        lib = _lib.lib()
        desc = ctx.desc
        link = ctx.cfg.get("first2_link")
        if link is not None and "grads" in link:
            # the layer on top of this one (ConvBNAct with cfg["first2"]) ran this layer's whole backward inside its own input-
            # gradient kernel (dsr_conv_dgrad_first_bwd) and left the results here; `dout` is a placeholder without storage
            dw, db = link.pop("grads")
            return None, dw, db, None, None
        opl = ctx.cfg.get("out_ps_link")
        got = opl.pop("dy", None) if opl is not None else None
        if opl is not None:
            opl.clear()
        if got is None:
            dout = dout.contiguous()     # (otherwise a placeholder without storage: see below)
        cout = desc.Cout
        n = x.shape[0]
        oh, ow = _out_hw(desc)
        cyp = r8(cout)
        need_partial = ctx.has_bias or prelu is not None
        if (not ctx.needs_input_grad[0] and ctx.needs_input_grad[1] and not ctx.ps and prelu is None
                and lib.dsr_conv_first_bwd_supported(C.byref(desc), ctx.act)):
            # first layer on an image (discriminator.py:22): activation mask, bias gradient and weight gradient in ONE
            # pass over dout and y -- g = dout * act'(y) is never materialised
            dw = torch.empty(ctx.wshape, dtype=torch.float32, device=x.device)
            db = torch.empty(cout, dtype=torch.float32, device=x.device) if ctx.has_bias else None
            wsz = lib.dsr_conv_first_bwd_workspace(C.byref(desc))
            ws = torch.empty(wsz, dtype=torch.uint8, device=x.device)
            w0 = getattr(ctx, "weight_ref", None)
            if (FIRST_BWD_RECOMPUTE and w0 is not None and w0.dtype == torch.float32 and w0.is_contiguous()
                    and _version(w0) == ctx.weight_version):
                # the activation output y is not read: the sign of the pre-activation is recomputed inside the pass from the
                # image and the layer's own weights (unchanged since the forward: same version) -- 1.07 GB less per pass at 512^2
                b0 = ctx.bias_ref
                check(_timed("wgrad", desc, lambda: lib.dsr_conv_first_bwd_recompute(
                    C.byref(desc), _ptr(x), _ptr(dout), _ptr(w0.detach()), _ptr(b0.detach() if b0 is not None else None), ctx.act,
                    float(ctx.cfg.get("slope", 0.0)), _ptr(dw), _ptr(db), _ptr(ws), wsz, _stream()), name="conv_first_bwd_kernel"))
            else:
                check(_timed("wgrad", desc, lambda: lib.dsr_conv_first_bwd(
                    C.byref(desc), _ptr(x), _ptr(dout), _ptr(y), ctx.act, float(ctx.cfg.get("slope", 0.0)), _ptr(dw), _ptr(db),
                    _ptr(ws), wsz, _stream()), name="conv_first_bwd_kernel"))
            return None, dw, db, None, None
        out_link = ctx.cfg.get("out_link")
        if got is not None:
            # the tail's input-gradient launch (dsr_conv_dgrad_ps) already produced the masked, un-shuffled gradient of this
            # layer's conv output and the partial sums; `dout` is a placeholder without storage
            dy, part, blocks = got
            db = dprelu = None
            if ctx.has_bias:
                db = torch.empty(cout, dtype=torch.float32, device=x.device)
                check(lib.dsr_pw_sum_rows(_ptr(part), blocks, 2 * cyp, 0, cout, 1.0, _ptr(db), 0, 1, _stream()))
            chan = torch.empty(cyp, dtype=torch.float32, device=x.device)
            check(lib.dsr_pw_sum_rows(_ptr(part), blocks, 2 * cyp, cyp, cyp, 1.0, _ptr(chan), 0, 1, _stream()))
            dprelu = torch.empty(1, dtype=torch.float32, device=x.device)
            check(lib.dsr_pw_sum_rows(_ptr(chan), cyp, 1, 0, 1, 1.0, _ptr(dprelu), 0, 0, _stream()))
        elif (out_link is not None and out_link.premasked and prelu is None and not ctx.ps
                and not (ctx.has_bias and ctx.needs_input_grad[2])):
            # the consumer of this layer's output already multiplied dout by act'(y) (ActLink): nothing left to do here
            out_link.premasked = False
            dy = dout
            db = dprelu = None
        elif ctx.act == ACT_NONE and not ctx.ps:
            dy = dout
            db = _colsum(dy, cout) if ctx.has_bias else None
            dprelu = None
        else:
            dy = torch.empty((n, oh, ow, cyp), dtype=x.dtype, device=x.device)
            p = n * oh * ow
            blocks, rpb = _reduce_blocks(p)
            part = torch.empty((blocks + _scr()) * 2 * cyp, dtype=torch.float32, device=x.device) if need_partial else None
            check(lib.dsr_pw_act_bwd(_dt(x), _ptr(dout), _ptr(y), _ptr(dy), n, oh, ow, cyp, y.shape[-1], int(ctx.ps),
                                     ctx.act, float(ctx.cfg.get("slope", 0.0)), _ptr(prelu), blocks, rpb, _ptr(part),
                                     _stream()))
            db = dprelu = None
            if ctx.has_bias:
                db = torch.empty(cout, dtype=torch.float32, device=x.device)
                check(lib.dsr_pw_sum_rows(_ptr(part), blocks, 2 * cyp, 0, cout, 1.0, _ptr(db), 0, 1, _stream()))
            if prelu is not None:
                chan = torch.empty(cyp, dtype=torch.float32, device=x.device)
                check(lib.dsr_pw_sum_rows(_ptr(part), blocks, 2 * cyp, cyp, cyp, 1.0, _ptr(chan), 0, 1, _stream()))
                dprelu = torch.empty(1, dtype=torch.float32, device=x.device)
                check(lib.dsr_pw_sum_rows(_ptr(chan), cyp, 1, 0, 1, 1.0, _ptr(dprelu), 0, 0, _stream()))
        dx, dw = _conv_backward(desc, x, dy, wd, ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.wshape,
                                 getattr(ctx, "weight_ref", None), in_link=ctx.cfg.get("in_link"))
        return dx, dw, db, dprelu, None


FIRST2_BACKWARD = os.environ.get("DSR_FIRST2_BACKWARD", "1") != "0"   # 0: the first two layers' backward as separate launches
DGRAD_BN = os.environ.get("DSR_DGRAD_BN_LINK", "1") != "0"             # 0: BatchNorm-backward sums always by their own reduce pass
DGRAD_PS = os.environ.get("DSR_DGRAD_PS_LINK", "1") != "0"             # 0: the PixelShuffle-PReLU backward always as its own pass


def _first2_backward(ctx, desc, x, dy, wd):
    """discriminator.py:25-29 in a step that needs no image gradient (the discriminator's own update): this layer's input
    gradient and the image layer's whole backward (mask, bias and weight gradient) as one launch, dsr_conv_dgrad_first_bwd --
    the gradient of the 64-channel activation between the two (1.07 GB at 512x512, batch 32) is never written.  Returns the
    placeholder to hand to autograd as that gradient (the image layer's gradients are left in the link its ConvAct node
    reads), or None when the pair does not qualify: then the caller takes the separate launches."""
    first2 = ctx.cfg.get("first2")
    if not FIRST2_BACKWARD or first2 is None or len(first2) < 5 or first2[4] is None or not ctx.needs_input_grad[0]:
        return None
    img, w0, b0, slope0, link = first2
    if (img.requires_grad or not w0.requires_grad or w0.dtype != torch.float32 or not w0.is_contiguous()
            or _version(w0) != getattr(ctx, "first2_w0_version", None) or (b0 is not None and not b0.requires_grad)):
        return None
    lib = _lib.lib()
    d0 = make_desc(img, w0.shape[0], 3, 3, 1, 1, PAD_ZERO, w0.shape[1])
    if not lib.dsr_conv_dgrad_first_bwd_supported(C.byref(d0), C.byref(desc), ACT_LEAKY):
        return None
    dw0 = torch.empty(tuple(w0.shape), dtype=torch.float32, device=x.device)
    db0 = torch.empty(w0.shape[0], dtype=torch.float32, device=x.device) if b0 is not None else None
    wsz = lib.dsr_conv_dgrad_first_bwd_workspace(C.byref(desc))
    ws = torch.empty(wsz, dtype=torch.uint8, device=x.device)
    check(_timed("dgrad", desc, lambda: lib.dsr_conv_dgrad_first_bwd(
        C.byref(d0), C.byref(desc), _ptr(dy), _ptr(wd), _ptr(img), _ptr(w0.detach()), _ptr(b0.detach() if b0 is not None else None),
        ACT_LEAKY, float(slope0), _ptr(dw0), _ptr(db0), _ptr(ws), wsz, _stream()), name="conv_dgrad_s2_kernel<first_bwd>"))
    link["grads"] = (dw0, db0)
    return torch.empty(1, dtype=x.dtype, device=x.device).expand(x.shape)


# ----------------------------------------------------------------------------- conv + BatchNorm + activation (+ residual)
class ConvBNAct(torch.autograd.Function):
    """out = act(BN(conv(x, W) + b)) [+ residual]   (train or eval mode BatchNorm2d).

    generator.py:14-25,71-74 (conv3x3 -> BN -> PReLU | + skip), discriminator.py:14-19
    (conv3x3 s1|s2 -> BN -> LeakyReLU), models/DIP/skip.py:54-88 (reflect conv -> BN -> LeakyReLU).
    The conv epilogue emits per-channel sum / sum-of-squares partial rows from its fp32
    accumulators; a tiny finalize kernel turns them into the affine map and updates the running
    statistics (momentum 0.1, unbiased variance) exactly like nn.BatchNorm2d."""

    @classmethod
    def apply(cls, *args):
        # forward() runs with grad mode off whatever the caller's mode is: note the caller's here ("infer" selects the
        # single-kernel inference epilogue, which saves nothing for a backward pass)
        cfg = dict(args[-1], infer=not torch.is_grad_enabled())
        return super().apply(*args[:-1], cfg)

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, running_mean, running_var, nbt, prelu, residual, cfg):
        _need_gpu(x)
        lib = _lib.lib()
        x = x.contiguous()
        cout, cin, kh, kw = weight.shape
        desc = make_desc(x, cout, kh, kw, cfg["stride"], cfg["pad"], cfg.get("pad_mode", PAD_ZERO), cin)
        oh, ow = _out_hw(desc)
        wf, wd = packed_weights(weight, desc, x.dtype)
        n = x.shape[0]
        cp = r8(cout)
        train = bool(cfg["train"])
        dev = x.device
        y = torch.empty((n, oh, ow, cp), dtype=x.dtype, device=dev)
        scale = torch.empty(cp, dtype=torch.float32, device=dev)
        shift = torch.empty(cp, dtype=torch.float32, device=dev)
        mean = torch.empty(cp, dtype=torch.float32, device=dev)
        rstd = torch.empty(cp, dtype=torch.float32, device=dev)
        count = n * oh * ow
        first2 = cfg.get("first2") if train else None
        if first2 is not None:
            # x is the (not yet computed) output of the image layer in front of this one: both convolutions run in ONE kernel
            # that recomputes that activation per tile in LDS (dsr_conv_first2_fwd) and writes it to x only when a backward
            # pass will read it (this layer's weight gradient); the forward itself never reads it back from HBM
            img, w0, b0, slope0 = first2[:4]
            ctx.first2_w0_version = _version(w0)
            d0 = make_desc(img, w0.shape[0], 3, 3, 1, 1, PAD_ZERO, w0.shape[1])
            wf0, _ = packed_weights(w0, d0, x.dtype)
            rows = lib.dsr_conv_first2_stats_rows(C.byref(d0))
            part = torch.empty((rows + _scr()) * 2 * cp, dtype=torch.float32, device=dev)
            keep = not cfg.get("infer", False)
            check(_timed("fwd", desc, lambda: lib.dsr_conv_first2_fwd(
                C.byref(d0), C.byref(desc), _ptr(img), _ptr(wf0), _ptr(b0.detach() if b0 is not None else None), float(slope0),
                _ptr(wf), _ptr(bias), _ptr(x) if keep else None, _ptr(y), _ptr(part), _stream()), name="conv_first2_kernel"))
        elif train:
            rows = lib.dsr_conv_stats_rows(C.byref(desc))
            part = torch.empty((rows + _scr()) * 2 * cp, dtype=torch.float32, device=dev)
            ep = Epilogue(ACT_NONE, 0.0, None, _ptr(bias), _ptr(part), 0, None)
            check(_timed("fwd", desc, lambda: lib.dsr_conv_fwd(C.byref(desc), _ptr(x), _ptr(wf), C.byref(ep), _ptr(y),
                                                                 _stream()), ep))
        if train:
            check(lib.dsr_pw_bn_finalize(_ptr(part), rows, cp, cout, cp, float(count), _ptr(gamma), _ptr(beta),
                                         _ptr(running_mean), _ptr(running_var), _ptr(nbt), BN_MOMENTUM, BN_EPS,
                                         int(cfg.get("bn_updates", 1)),
                                         _ptr(scale), _ptr(shift), _ptr(mean), _ptr(rstd), _stream()))
            bump(running_mean)       # rewritten through their raw pointers: whatever is keyed on them (the inference
            bump(running_var)        # affine map kept with running_mean, below) must see a new version
        else:
            infer = cfg.get("infer", False) and lib.dsr_conv_fwd_affine_supported(C.byref(desc))
            # inference: the affine map of an eval-mode BatchNorm changes only when its four tensors do -- keep it with the
            # running mean, keyed by their version counters (33 launches of ~3 us per x8 forward otherwise)
            # (_version(p) = torch's counter + the counter bump() advances when a HIP launch rewrites p through its raw
            #  pointer -- FusedAdam for gamma / beta, the train-mode finalize below for the running statistics -- + the address)
            key = (_version(gamma), _version(beta), _version(running_mean), _version(running_var), str(dev))
            capturing = torch.cuda.is_current_stream_capturing()
            # a capture never takes the kept map: the graph would replay yesterday's scale / shift whatever the statistics
            # are by then -- inside a graph the affine launch is part of every replay
            hit = getattr(running_mean, "_dsr_affine", None) if (infer and not capturing) else None
            if hit is not None and hit[0] == key:
                scale, shift = hit[1], hit[2]
                if hit[4] != torch.cuda.current_stream(dev).cuda_stream:
                    torch.cuda.current_stream(dev).wait_event(hit[3])      # computed on another stream
            else:
                check(lib.dsr_pw_bn_eval_affine(_ptr(gamma), _ptr(beta), _ptr(running_mean), _ptr(running_var), BN_EPS,
                                                cout, cp, _ptr(scale), _ptr(shift), _ptr(mean), _ptr(rstd), _stream()))
                if infer and not capturing:
                    ev = torch.cuda.Event()
                    ev.record()
                    running_mean._dsr_affine = (key, scale, shift, ev, torch.cuda.current_stream(dev).cuda_stream)
            if infer:
                # inference (eval_GAN.py:44,94; called under torch.no_grad(), see apply() below): eval-mode BatchNorm
                # is a fixed per-channel affine map, so it, the activation and the skip connection all ride in the conv
                # epilogue -- one kernel, one pass
                res = residual.contiguous() if residual is not None else None
                ep = Epilogue(cfg.get("act", ACT_NONE), float(cfg.get("slope", 0.0)), _ptr(prelu), _ptr(bias), None, 0,
                              None, _ptr(scale), _ptr(shift), _ptr(res))
                check(_timed("fwd", desc, lambda: lib.dsr_conv_fwd(C.byref(desc), _ptr(x), _ptr(wf), C.byref(ep), _ptr(y),
                                                                     _stream()), ep))
                return (y, x) if cfg.get("carry_input", False) else y
            ep = Epilogue(ACT_NONE, 0.0, None, _ptr(bias), None, 0, None)
            check(_timed("fwd", desc, lambda: lib.dsr_conv_fwd(C.byref(desc), _ptr(x), _ptr(wf), C.byref(ep), _ptr(y),
                                                                 _stream()), ep))
        act = cfg.get("act", ACT_NONE)
        out = torch.empty_like(y)
        res = residual.contiguous() if residual is not None else None
        check(lib.dsr_pw_bn_act_fwd(_dt(x), _ptr(y), _ptr(scale), _ptr(shift), _ptr(res), _ptr(out), count, cp, act,
                                    float(cfg.get("slope", 0.0)), _ptr(prelu), _stream()))
        ctx.desc, ctx.cfg, ctx.act, ctx.train, ctx.count = desc, cfg, act, train, count
        bol = cfg.get("bn_out_link")
        if bol is not None:
            # the layer that consumes `out` may form this layer's BatchNorm-backward sums inside its input-gradient launch
            bol.clear()
            if train and residual is None and prelu is None and not cfg.get("infer", False):
                bol.update(y=y, scale=scale, shift=shift, act=act, slope=float(cfg.get("slope", 0.0)))
        ctx.wshape = tuple(weight.shape)
        ctx.weight_ref = weight
        ctx.has_res = residual is not None
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x, y, wd, scale, shift, mean, rstd,
                              prelu if prelu is not None else torch.empty(0, device=dev))
        if cfg.get("carry_input", False):
            # the input comes back as a second output: a residual block hands THAT to its skip connection, so both gradients
            # of the block input arrive at this node and are summed in its dgrad epilogue (dsr_conv_dgrad_add) instead of by
            # an elementwise pass of autograd's
            return out, x
        return out

    @staticmethod
    def backward(ctx, dout, dcarry=None):
        x, y, wd, scale, shift, mean, rstd, prelu = ctx.saved_tensors
        prelu = prelu if prelu.numel() else None
        lib = _lib.lib()
        desc = ctx.desc
        dout = dout.contiguous()
        cout = desc.Cout
        cp = y.shape[-1]
        dev = x.device
        p = ctx.count
        slope = float(ctx.cfg.get("slope", 0.0))
        c1 = torch.empty(cp, dtype=torch.float32, device=dev)
        c2 = torch.empty(cp, dtype=torch.float32, device=dev)
        dgamma = torch.empty(cout, dtype=torch.float32, device=dev)
        dbeta = torch.empty(cout, dtype=torch.float32, device=dev)
        dprelu = torch.empty(1, dtype=torch.float32, device=dev) if prelu is not None else None
        bol = ctx.cfg.get("bn_out_link")
        got = bol.pop("part", None) if bol is not None else None
        if bol is not None:
            bol.clear()
        if got is not None and got[2] == dout.data_ptr():
            # the launch that produced `dout` (the next layer's input gradient, dsr_conv_dgrad_bn) formed the two sums already
            part, blocks = got[0], got[1]
        else:
            blocks, rpb = _bn_bwd_blocks(p, ctx.act)
            part = torch.empty((blocks + _scr()) * 3 * cp, dtype=torch.float32, device=dev)
            check(lib.dsr_pw_bn_act_bwd_reduce(_dt(x), _ptr(dout), _ptr(y), _ptr(scale), _ptr(shift), _ptr(mean),
                                               _ptr(rstd), p, cp, blocks, rpb, ctx.act, slope, _ptr(prelu), _ptr(part),
                                               _stream()))
        check(lib.dsr_pw_bn_bwd_finalize(_ptr(part), blocks, cout, cp, float(p), _ptr(mean), _ptr(rstd), _ptr(dgamma),
                                         _ptr(dbeta), _ptr(dprelu), _ptr(c1), _ptr(c2), _stream()))
        dy = torch.empty_like(y)
        check(lib.dsr_pw_bn_act_bwd_apply(_dt(x), _ptr(dout), _ptr(y), _ptr(scale), _ptr(shift), _ptr(mean), _ptr(rstd),
                                          _ptr(c1), _ptr(c2), _ptr(dy), p, cp, ctx.act, slope, _ptr(prelu),
                                          int(ctx.train), _stream()))
        fused = _first2_backward(ctx, desc, x, dy, wd) if dcarry is None else None
        if fused is not None:
            # the input gradient (of the image layer's activation) was consumed where it was formed: autograd gets a
            # placeholder of the right shape without storage, and the image layer's ConvAct node finds its gradients in the link
            dx = fused
            _, dw = _conv_backward(desc, x, dy, wd, False, ctx.needs_input_grad[1], ctx.wshape, getattr(ctx, "weight_ref", None))
        else:
            dx, dw = _conv_backward(desc, x, dy, wd, ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.wshape,
                                     getattr(ctx, "weight_ref", None), dcarry, bn_link=ctx.cfg.get("bn_in_link"))
        db = None
        if ctx.has_bias and not ctx.train:
            # a bias in front of a train-mode BatchNorm has an analytically zero gradient (the reference holds ~1e-9
            # rounding noise there): no gradient is produced for it at all -- Adam leaves such a parameter exactly where
            # a zero gradient would (m = v = 0 => no update), and 47 zero-fill launches per GAN step disappear.
            # In eval mode it is the column sum of dy.
            db = _colsum(dy, cout)
        dres = dout if ctx.has_res else None
        return dx, dw, db, dgamma, dbeta, None, None, None, dprelu, dres, None


# ----------------------------------------------------------------------------- last layer: conv + act -> fp32 NCHW
class ConvOutNCHW(torch.autograd.Function):
    """fp32 NCHW output of the last conv (+Tanh: generator.py:78-80; +Sigmoid: models/DIP/skip.py:92-94)."""

    @staticmethod
    def forward(ctx, x, weight, bias, cfg):
        _need_gpu(x)
        x = x.contiguous()
        cout, cin, kh, kw = weight.shape
        desc = make_desc(x, cout, kh, kw, cfg["stride"], cfg["pad"], cfg.get("pad_mode", PAD_ZERO), cin)
        oh, ow = _out_hw(desc)
        wf, wd = packed_weights(weight, desc, x.dtype)
        out = torch.empty((x.shape[0], cout, oh, ow), dtype=torch.float32, device=x.device)
        act = cfg.get("act", ACT_NONE)
        ep = Epilogue(act, 0.0, None, _ptr(bias), None, 0, _ptr(out))
        check(_timed("fwd", desc, lambda: _lib.lib().dsr_conv_fwd(C.byref(desc), _ptr(x), _ptr(wf), C.byref(ep), None,
                                                                    _stream()), ep))
        ctx.desc, ctx.act = desc, act
        ctx.ps_link = cfg.get("in_ps_link")
        ctx.wshape = tuple(weight.shape)
        ctx.weight_ref = weight
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x, out, wd)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, out, wd = ctx.saved_tensors
        desc = ctx.desc
        dout = dout.contiguous().float()
        n, c, h, w = out.shape
        dy = torch.empty((n, h, w, r8(c)), dtype=x.dtype, device=x.device)
        check(_lib.lib().dsr_pw_act_bwd_nchw(_dt(x), _ptr(dout), _ptr(out), _ptr(dy), n, c, h, w, r8(c), ctx.act,
                                             _stream()))
        lib = _lib.lib()
        link = getattr(ctx, "ps_link", None)
        if (link is not None and "prelu" in link and DGRAD_PS and ctx.needs_input_grad[0]
                and lib.dsr_conv_dgrad_ps_supported(C.byref(desc))):
            # x is PReLU(PixelShuffle(conv)) (generator.py:37-39 in front of :78): that activation's backward rides in this
            # layer's input-gradient launch (dsr_conv_dgrad_ps) -- the 64-channel gradient at the high resolution is never
            # written; the shuffle conv's node finds the un-shuffled, masked gradient and the partial sums in the link
            n_, h_, w_, _ = x.shape
            rows = lib.dsr_conv_dgrad_ps_rows(C.byref(desc))
            dyu = torch.empty((n_, h_ // 2, w_ // 2, 256), dtype=x.dtype, device=x.device)
            part = torch.empty((rows + _scr()) * 2 * 256, dtype=torch.float32, device=x.device)
            check(_timed("dgrad", desc, lambda: lib.dsr_conv_dgrad_ps(C.byref(desc), _ptr(dy), _ptr(wd), _ptr(x), _ptr(link["prelu"]),
                                                                       _ptr(dyu), _ptr(part), _stream()),
                         name="conv_dgrad_toeplitz9_kernel<ps>"))
            link["dy"] = (dyu, part, rows)
            dx = torch.empty(1, dtype=x.dtype, device=x.device).expand(x.shape)       # placeholder without storage
            _, dw = _conv_backward(desc, x, dy, wd, False, ctx.needs_input_grad[1], ctx.wshape, getattr(ctx, "weight_ref", None))
        else:
            dx, dw = _conv_backward(desc, x, dy, wd, ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.wshape,
                                     getattr(ctx, "weight_ref", None))
        db = _colsum(dy, c) if ctx.has_bias else None
        return dx, dw, db, None


# ----------------------------------------------------------------------------- losses
def axpby(x, y=None, a=1.0, b=1.0, g=None):
    """(a * x + b * y) * g on fp32 tensors (y optional, g an optional 0-dim / 1-element device tensor): dsr_pw_axpby_f32."""
    x = x.contiguous()
    out = torch.empty_like(x)
    if y is not None:
        y = y.contiguous()
        assert y.shape == x.shape and y.dtype == torch.float32
    if g is not None:
        g = g.contiguous().float()
    check(_lib.lib().dsr_pw_axpby_f32(_ptr(x), _ptr(y), float(a), float(b), _ptr(g), _ptr(out), x.numel(), _stream()))
    return out


class DiffLoss(torch.autograd.Function):
    """mean |a-b| (mode 0, BASELINE config 2) or mean (a-b)^2 (mode 1, nn.MSELoss: DIP.py:26,65;
    utils/GAN.py:74,90) over fp32 tensors.  Both arguments are differentiable (nn.MSELoss / nn.L1Loss are):
    d/d target = - d/d pred."""

    @staticmethod
    def forward(ctx, pred, target, mode):
        _need_gpu(pred)
        pred = pred.contiguous().float()
        target = target.contiguous().float()
        if pred.shape != target.shape:
            raise RuntimeError(f"loss operands differ in shape: {tuple(pred.shape)} vs {tuple(target.shape)}")
        n = pred.numel()
        blocks = max(1, min(1024, (n + 1023) // 1024))
        part = torch.empty(blocks, dtype=torch.float32, device=pred.device)
        grad = torch.empty_like(pred)
        lib = _lib.lib()
        check(lib.dsr_pw_diff_loss(_ptr(pred), _ptr(target), _ptr(grad), n, mode, _ptr(part), blocks, _stream()))
        loss = torch.empty(1, dtype=torch.float32, device=pred.device)
        check(lib.dsr_pw_sum_rows(_ptr(part), blocks, 1, 0, 1, 1.0 / n, _ptr(loss), 0, 0, _stream()))
        ctx.save_for_backward(grad)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        gp = axpby(grad, None, 1.0, 0.0, g) if ctx.needs_input_grad[0] else None
        gt = axpby(grad, None, -1.0, 0.0, g) if ctx.needs_input_grad[1] else None
        return gp, gt, None


def l1_loss(pred, target):
    return DiffLoss.apply(pred, target, 0)


def mse_loss(pred, target):
    return DiffLoss.apply(pred, target, 1)


class BCEConst(torch.autograd.Function):
    """nn.BCELoss()(p, full_like(p, target)) with the log clamp at -100 (utils/GAN.py:96-105)."""

    @staticmethod
    def forward(ctx, p, target):
        _need_gpu(p)
        p = p.contiguous().float()
        loss = torch.empty(1, dtype=torch.float32, device=p.device)
        grad = torch.empty_like(p)
        check(_lib.lib().dsr_pw_bce_const(_ptr(p), p.numel(), float(target), _ptr(loss), _ptr(grad), 0, _stream()))
        ctx.save_for_backward(grad)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return axpby(grad, None, 1.0, 0.0, g), None


def bce_const(p, target):
    return BCEConst.apply(p, target)


class AddScalars(torch.autograd.Function):
    """a + b for two scalar losses (utils/GAN.py:105 BCE(real,1) + BCE(fake,0); :122 content + adversarial)."""

    @staticmethod
    def forward(ctx, a, b):
        return axpby(a.reshape(1), b.reshape(1), 1.0, 1.0).reshape(())

    @staticmethod
    def backward(ctx, g):
        return g, g


def add_losses(a, b):
    return AddScalars.apply(a, b)


class ScaleLoss(torch.autograd.Function):
    """s * loss for a host constant s (the static loss scale of the fp16 DIP path)."""

    @staticmethod
    def forward(ctx, loss, s):
        ctx.s = float(s)
        return axpby(loss.reshape(1), None, ctx.s, 0.0).reshape(())

    @staticmethod
    def backward(ctx, g):
        return axpby(g.reshape(1), None, ctx.s, 0.0).reshape(()), None


def scale_loss(loss, s):
    return loss if s == 1.0 else ScaleLoss.apply(loss, s)


# ----------------------------------------------------------------------------- discriminator dense head
_shadow_cache = {}


def shadow16(weight, dtype):
    """16-bit copy of an fp32 matrix in its own layout, refreshed when the parameter changes."""
    key = (id(weight), dtype, tuple(weight.shape))
    ver = _version(weight)
    hit = _shadow_cache.get(key)
    same = hit is not None and hit[2]() is weight
    if same and hit[0] == ver:
        return hit[1]
    w = weight.detach().contiguous()
    out = hit[1] if same else torch.empty(w.shape, dtype=dtype, device=w.device)
    check(_lib.lib().dsr_cast16(_dt(out), _ptr(w), _ptr(out), w.numel(), _stream()))
    _shadow_cache[key] = (ver, out, weakref.ref(weight))
    return out


def shadow_for_update(weight):
    """The bf16 shadow tensor of `weight` if one exists (so the optimiser can refresh it in its own pass)."""
    hit = _shadow_cache.get((id(weight), torch.bfloat16, tuple(weight.shape)))
    if hit is not None and hit[2]() is weight and weight.is_contiguous():
        return hit[1]
    return None


def mark_shadow_current(weight):
    key = (id(weight), torch.bfloat16, tuple(weight.shape))
    hit = _shadow_cache.get(key)
    if hit is not None:
        _shadow_cache[key] = (_version(weight), hit[1], hit[2])


def _awaits_allreduce(param):
    """True when a data-parallel run will average `param`.grad with a collective after backward (dist.GradSync) -- such a
    gradient has to exist as a tensor; the dense head's matrix is exempt when its factors are gathered instead."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return False
    return not getattr(param, "_dsr_grad_global", False)


def dp_world_for(param):
    """World size when `param`'s gradient is produced already averaged over the ranks by its own backward
    (`param._dsr_grad_global`, set by dist.GradSync.attach for the dense head's big matrix), else 0."""
    if not getattr(param, "_dsr_grad_global", False):
        return 0
    import torch.distributed as dist
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 0


def _gather_factors(xt, dyt, world):
    """Asynchronous all-gather of the rank-local wgrad factors; returns the [R][...] buffers and the work handles."""
    import torch.distributed as dist
    xt_all = torch.empty((world,) + tuple(xt.shape), dtype=xt.dtype, device=xt.device)
    dyt_all = torch.empty((world,) + tuple(dyt.shape), dtype=dyt.dtype, device=dyt.device)
    if dist.get_backend() == "gloo":        # (rehearsal on one GPU) no all_gather_into_tensor for device tensors
        works = [dist.all_gather(list(xt_all.unbind(0)), xt, async_op=True),
                 dist.all_gather(list(dyt_all.unbind(0)), dyt, async_op=True)]
    else:
        works = [dist.all_gather_into_tensor(xt_all, xt, async_op=True),
                 dist.all_gather_into_tensor(dyt_all, dyt, async_op=True)]
    return xt_all, dyt_all, works


class DenseHead(torch.autograd.Function):
    """sigmoid(Linear(1024,1)(leaky_relu(Linear(K,1024)(flatten_CHW(x)), 0.2)))  -- discriminator.py:65-72.

    x is the NHWC 16-bit output of the last conv block; the C,H,W flatten order of ``x.view(N,-1)`` on an
    NCHW tensor (:65) is reproduced by a small transposing copy, so dense1.weight keeps the reference layout."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, c):
        _need_gpu(x)
        lib = _lib.lib()
        x = x.contiguous()
        n, h, w, cp = x.shape
        hw = h * w
        k = c * hw
        o = w1.shape[0]
        if w1.shape[1] != k:
            raise RuntimeError(f"dense1 expects {w1.shape[1]} features, the conv stack produced {k}")   # torch raises too
        dev = x.device
        st = _stream()
        flat = torch.empty((n, k), dtype=x.dtype, device=dev)
        check(lib.dsr_flatten(_dt(x), _ptr(x), _ptr(flat), n, hw, c, cp, 0, 0, st))
        w16 = shadow16(w1, x.dtype)
        wsz = lib.dsr_linear_fwd_workspace(n, k, o)
        ws = torch.empty(wsz, dtype=torch.uint8, device=dev)
        h1 = torch.empty((n, o), dtype=torch.float32, device=dev)
        check(lib.dsr_linear_fwd(_dt(x), _ptr(flat), _ptr(w16), _ptr(b1), ACT_LEAKY, 0.2, _ptr(h1), n, k, o, _ptr(ws),
                                 wsz, st))
        out = torch.empty((n, 1), dtype=torch.float32, device=dev)
        check(lib.dsr_dense2_fwd(_ptr(h1), _ptr(w2), _ptr(b2), n, o, _ptr(out), st))
        ctx.c = c
        ctx.dp_world = dp_world_for(w1)
        ctx.w1 = w1                     # (the Parameter itself: backward may hand its gradient over in factored form)
        ctx.save_for_backward(x, h1, out, w16, w2)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, h1, out, w16, w2 = ctx.saved_tensors
        lib = _lib.lib()
        n, h, w, cp = x.shape
        hw, c = h * w, ctx.c
        k = c * hw
        o = h1.shape[1]
        dev = x.device
        st = _stream()
        bp = 32 if n <= 32 else 64
        dout = dout.contiguous().float()
        dw2 = torch.empty((1, o), dtype=torch.float32, device=dev)
        db2 = torch.empty(1, dtype=torch.float32, device=dev)
        db1 = torch.empty(o, dtype=torch.float32, device=dev)
        dy16 = torch.empty((n, o), dtype=x.dtype, device=dev)
        dyt16 = torch.empty((o, bp), dtype=x.dtype, device=dev)
        check(lib.dsr_dense2_bwd(_dt(x), _ptr(dout), _ptr(out), _ptr(h1), _ptr(w2), n, o, bp, 0.2, _ptr(dw2), _ptr(db2),
                                 _ptr(db1), _ptr(dy16), _ptr(dyt16), st))
        dx = dw1 = None
        gather = None
        if ctx.needs_input_grad[1]:
            xt = torch.empty((k, bp), dtype=x.dtype, device=dev)
            check(lib.dsr_flatten(_dt(x), _ptr(x), _ptr(xt), n, hw, c, cp, bp, 1, st))
            if ctx.dp_world >= 1:     # (1 only under DSR_DIST_FORCE=1: the RCCL path rehearsed on a single rank)
                # data parallel: the averaged dW1 = (1/R) sum_r dyT_r xT_r is formed from the all-gathered rank-local factors
                # (67 MB + 128 KB per rank) instead of all-reducing the 2.1 GB gradient; dist.GradSync skips this tensor
                gather = _gather_factors(xt, dyt16, ctx.dp_world)
        if ctx.needs_input_grad[0]:          # issued before the gather is waited for: it runs under the exchange
            dflat = torch.empty((n, k), dtype=x.dtype, device=dev)
            check(lib.dsr_linear_dgrad(_dt(x), _ptr(dy16), _ptr(w16), _ptr(dflat), n, o, k, st))
            dx = torch.empty_like(x)
            check(lib.dsr_flatten(_dt(x), _ptr(dflat), _ptr(dx), n, hw, c, cp, 0, 2, st))
        if ctx.needs_input_grad[1]:
            works = ()
            if gather is not None:
                xt, dyt16, works = gather
            ranks = max(ctx.dp_world, 1) if gather is not None else 1
            fac = GradFactors(_dt(x), dyt16, xt, bp, o, k, ranks, 1.0 / ranks, works)
            if getattr(ctx.w1, "_dsr_defer_wgrad", False) and k % 64 == 0 and not _awaits_allreduce(ctx.w1):
                # optim.FusedAdam(fuse_dense_head=True): dW1 = dyT x is a rank-(64 R) product -- hand the two factors to
                # the optimiser, whose dsr_linear_wgrad_adam launch forms each tile of it in registers and applies Adam
                # there; the 2.1 GB gradient (config 3) is neither written nor read back
                pending = getattr(ctx.w1, "_dsr_grad_factors", None)
                if pending is None:
                    pending = ctx.w1._dsr_grad_factors = []
                pending.append(fac)
            else:
                dw1 = fac.materialize()
        return dx, dw1, db1, dw2, db2, None


class GradFactors:
    """The gradient of a Linear weight as its two 16-bit factors: dW[o][k] = scale * sum_r sum_b dyT[r][o][b] xT[r][k][b]
    (R rank-local pairs, all-gathered under data parallelism -- `works` are the pending gathers)."""

    def __init__(self, dt, dyt, xt, bp, o, k, ranks, scale, works):
        self.dt, self.dyt, self.xt, self.bp, self.o, self.k = dt, dyt, xt, bp, o, k
        self.ranks, self.scale, self.works = ranks, scale, tuple(works)

    def wait(self):
        for wk in self.works:
            wk.wait()
        self.works = ()

    def materialize(self):
        """The fp32 [o][k] gradient (what Tensor.grad would have held)."""
        lib = _lib.lib()
        self.wait()
        dw = torch.empty((self.o, self.k), dtype=torch.float32, device=self.xt.device)
        if self.ranks > 1 or self.scale != 1.0:
            check(lib.dsr_linear_wgrad_gathered(self.dt, _ptr(self.dyt), _ptr(self.xt), _ptr(dw), self.bp, self.o, self.k,
                                                self.ranks, self.scale, _stream()))
        else:
            check(lib.dsr_linear_wgrad(self.dt, _ptr(self.dyt), _ptr(self.xt), _ptr(dw), self.bp, self.o, self.k, _stream()))
        return dw


# ----------------------------------------------------------------------------- pooling / resampling
class MaxPool2(torch.autograd.Function):
    """nn.MaxPool2d(2, 2) on NHWC (VGG19 trunk, utils/GAN.py:24-47)."""

    @staticmethod
    def forward(ctx, x, *rest):
        in_link = rest[0] if rest else None               # (optional second argument: the ActLink of the ReLU that produced x)
        ctx.nin = 1 + len(rest)
        x = x.contiguous()
        n, h, w, cp = x.shape
        y = torch.empty((n, h // 2, w // 2, cp), dtype=x.dtype, device=x.device)
        check(_lib.lib().dsr_maxpool2_fwd(_dt(x), _ptr(x), _ptr(y), n, h, w, cp, _stream()))
        ctx.in_link = in_link
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        n, h, w, cp = x.shape
        dx = torch.empty_like(x)
        link = ctx.in_link
        if link is not None and link.act == ACT_RELU:
            # x is a ReLU output with this pool as its only consumer: the ReLU's backward rides in the routing (ActLink)
            check(_lib.lib().dsr_maxpool2_relu_bwd(_dt(x), _ptr(x), _ptr(dy.contiguous()), _ptr(dx), n, h, w, cp, _stream()))
            link.premasked = True
        else:
            check(_lib.lib().dsr_maxpool2_bwd(_dt(x), _ptr(x), _ptr(dy.contiguous()), _ptr(dx), n, h, w, cp, _stream()))
        return (dx, None) if ctx.nin == 2 else dx


class AvgPool2(torch.autograd.Function):
    """nn.AvgPool2d(2, 2) on NHWC: conv(..., downsample_mode='avg') of models/DIP/utils.py:86-94."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        n, h, w, cp = x.shape
        y = torch.empty((n, h // 2, w // 2, cp), dtype=x.dtype, device=x.device)
        check(_lib.lib().dsr_avgpool2_fwd(_dt(x), _ptr(x), _ptr(y), n, h, w, cp, _stream()))
        ctx.shape = (n, h, w, cp)
        return y

    @staticmethod
    def backward(ctx, dy):
        n, h, w, cp = ctx.shape
        dy = dy.contiguous()
        dx = torch.empty((n, h, w, cp), dtype=dy.dtype, device=dy.device)
        check(_lib.lib().dsr_avgpool2_bwd(_dt(dy), _ptr(dy), _ptr(dx), n, h, w, cp, _stream()))
        return dx


class Nearest2x(torch.autograd.Function):
    """nn.Upsample(scale_factor=2, mode='nearest') on NHWC (models/DIP/skip.py:77, skip()'s default mode)."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        n, h, w, cp = x.shape
        y = torch.empty((n, 2 * h, 2 * w, cp), dtype=x.dtype, device=x.device)
        check(_lib.lib().dsr_nearest2x_fwd(_dt(x), _ptr(x), _ptr(y), n, h, w, cp, _stream()))
        ctx.shape = (n, h, w, cp)
        return y

    @staticmethod
    def backward(ctx, dy):
        n, h, w, cp = ctx.shape
        dy = dy.contiguous()
        dx = torch.empty((n, h, w, cp), dtype=dy.dtype, device=dy.device)
        check(_lib.lib().dsr_nearest2x_bwd(_dt(dy), _ptr(dy), _ptr(dx), n, h, w, cp, _stream()))
        return dx


class Bilinear2x(torch.autograd.Function):
    """nn.Upsample(scale_factor=2, mode='bilinear') (align_corners=False) on NHWC (models/DIP/skip.py:77)."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        n, h, w, cp = x.shape
        y = torch.empty((n, 2 * h, 2 * w, cp), dtype=x.dtype, device=x.device)
        check(_lib.lib().dsr_bilinear2x_fwd(_dt(x), _ptr(x), _ptr(y), n, h, w, cp, _stream()))
        ctx.shape = (n, h, w, cp)
        return y

    @staticmethod
    def backward(ctx, dy):
        n, h, w, cp = ctx.shape
        dy = dy.contiguous()
        dx = torch.empty((n, h, w, cp), dtype=dy.dtype, device=dy.device)
        check(_lib.lib().dsr_bilinear2x_bwd(_dt(dy), _ptr(dy), _ptr(dx), n, h, w, cp, _stream()))
        return dx


class ResizeNorm(torch.autograd.Function):
    """torchvision ImageClassification preset on an fp32 NCHW batch -> NHWC 16-bit [N, crop, crop, 8].

    ``tab`` is a ResampleTables object (utils/GAN.py mirror) holding the device-resident forward and
    transposed weight tables of the separable antialiased resize + centre crop."""

    @staticmethod
    def forward(ctx, img, tab, dtype):
        _need_gpu(img)
        img = img.contiguous().float()
        n, c, h, w = img.shape
        assert (h, w) == (tab.in_h, tab.in_w), ((h, w), (tab.in_h, tab.in_w))
        out = torch.empty((n, tab.out_h, tab.out_w, 8), dtype=dtype, device=img.device)
        check(_lib.lib().dsr_resize_norm_fwd(_dt(out), _ptr(img), _ptr(out), n, c, h, w, tab.out_h, tab.out_w,
                                             _ptr(tab.ys), _ptr(tab.yc), _ptr(tab.yw), _ptr(tab.xs), _ptr(tab.xc),
                                             _ptr(tab.xw), tab.kt, tab.mean_c, tab.std_c, _stream()))
        ctx.tab, ctx.shape = tab, (n, c, h, w)
        return out

    @staticmethod
    def backward(ctx, dout):
        tab = ctx.tab
        n, c, h, w = ctx.shape
        dout = dout.contiguous()
        dimg = torch.empty((n, c, h, w), dtype=torch.float32, device=dout.device)
        check(_lib.lib().dsr_resize_norm_bwd(_dt(dout), _ptr(dout), _ptr(dimg), n, c, h, w, tab.out_h, tab.out_w,
                                             _ptr(tab.tys), _ptr(tab.tyc), _ptr(tab.tyw), _ptr(tab.txs), _ptr(tab.txc),
                                             _ptr(tab.txw), tab.kt, tab.std_c, _stream()))
        return dimg, None, None


def _box_copy(src, dst, n, bh, bw, c, sy0, sx0, cs0, dy0, dx0, cd0):
    check(_lib.lib().dsr_box_copy(_ptr(src), _ptr(dst), n, bh, bw, c, src.shape[1], src.shape[2], src.shape[3], sy0, sx0,
                                  cs0, dst.shape[1], dst.shape[2], dst.shape[3], dy0, dx0, cd0, _stream()))


class ConcatCrop(torch.autograd.Function):
    """torch.cat([a, b], dim=1) after centre-cropping both to the smaller H, W (models/DIP/utils.py:18-38)."""

    @staticmethod
    def forward(ctx, a, b, ca, cb):
        a, b = a.contiguous(), b.contiguous()
        n = a.shape[0]
        h, w = min(a.shape[1], b.shape[1]), min(a.shape[2], b.shape[2])
        out = torch.zeros((n, h, w, r8(ca + cb)), dtype=a.dtype, device=a.device)
        offs = []
        c0 = 0
        for t, c in ((a, ca), (b, cb)):
            d2, d3 = (t.shape[1] - h) // 2, (t.shape[2] - w) // 2
            _box_copy(t, out, n, h, w, c, d2, d3, 0, 0, 0, c0)
            offs.append((d2, d3, c0, c, tuple(t.shape)))
            c0 += c
        ctx.offs, ctx.hw = offs, (h, w)
        return out

    @staticmethod
    def backward(ctx, dout):
        dout = dout.contiguous()
        n = dout.shape[0]
        h, w = ctx.hw
        grads = []
        for d2, d3, c0, c, shape in ctx.offs:
            g = torch.zeros(shape, dtype=dout.dtype, device=dout.device)
            _box_copy(dout, g, n, h, w, c, 0, 0, c0, d2, d3, 0)
            grads.append(g)
        return grads[0], grads[1], None, None


class BNAct(torch.autograd.Function):
    """act(BatchNorm2d(x)) on a tensor that does not come straight out of a conv (DIP: BN after Concat,
    models/DIP/skip.py:51)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, nbt, c, cfg):
        _need_gpu(x)
        lib = _lib.lib()
        x = x.contiguous()
        cp = x.shape[-1]
        p = x.numel() // cp
        dev = x.device
        train = bool(cfg["train"])
        scale = torch.empty(cp, dtype=torch.float32, device=dev)
        shift = torch.empty(cp, dtype=torch.float32, device=dev)
        mean = torch.empty(cp, dtype=torch.float32, device=dev)
        rstd = torch.empty(cp, dtype=torch.float32, device=dev)
        if train:
            blocks, rpb = _reduce_blocks(p)
            part = torch.empty((blocks + _scr()) * 2 * cp, dtype=torch.float32, device=dev)
            check(lib.dsr_pw_channel_stats(_dt(x), _ptr(x), p, cp, blocks, rpb, _ptr(part), _stream()))
            check(lib.dsr_pw_bn_finalize(_ptr(part), blocks, cp, c, cp, float(p), _ptr(gamma), _ptr(beta),
                                         _ptr(running_mean), _ptr(running_var), _ptr(nbt), BN_MOMENTUM, BN_EPS,
                                         int(cfg.get("bn_updates", 1)),
                                         _ptr(scale), _ptr(shift), _ptr(mean), _ptr(rstd), _stream()))
            bump(running_mean)
            bump(running_var)
        else:
            check(lib.dsr_pw_bn_eval_affine(_ptr(gamma), _ptr(beta), _ptr(running_mean), _ptr(running_var), BN_EPS, c,
                                            cp, _ptr(scale), _ptr(shift), _ptr(mean), _ptr(rstd), _stream()))
        act = cfg.get("act", ACT_NONE)
        out = torch.empty_like(x)
        check(lib.dsr_pw_bn_act_fwd(_dt(x), _ptr(x), _ptr(scale), _ptr(shift), None, _ptr(out), p, cp, act,
                                    float(cfg.get("slope", 0.0)), None, _stream()))
        ctx.cfg, ctx.act, ctx.train, ctx.p, ctx.c = cfg, act, train, p, c
        ctx.save_for_backward(x, scale, shift, mean, rstd)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, scale, shift, mean, rstd = ctx.saved_tensors
        lib = _lib.lib()
        dout = dout.contiguous()
        cp = x.shape[-1]
        dev = x.device
        p, c = ctx.p, ctx.c
        slope = float(ctx.cfg.get("slope", 0.0))
        c1 = torch.empty(cp, dtype=torch.float32, device=dev)
        c2 = torch.empty(cp, dtype=torch.float32, device=dev)
        dgamma = torch.empty(c, dtype=torch.float32, device=dev)
        dbeta = torch.empty(c, dtype=torch.float32, device=dev)
        blocks, rpb = _bn_bwd_blocks(p, ctx.act)
        part = torch.empty((blocks + _scr()) * 3 * cp, dtype=torch.float32, device=dev)
        check(lib.dsr_pw_bn_act_bwd_reduce(_dt(x), _ptr(dout), _ptr(x), _ptr(scale), _ptr(shift), _ptr(mean), _ptr(rstd),
                                           p, cp, blocks, rpb, ctx.act, slope, None, _ptr(part), _stream()))
        check(lib.dsr_pw_bn_bwd_finalize(_ptr(part), blocks, c, cp, float(p), _ptr(mean), _ptr(rstd), _ptr(dgamma),
                                         _ptr(dbeta), None, _ptr(c1), _ptr(c2), _stream()))
        dx = torch.empty_like(x)
        check(lib.dsr_pw_bn_act_bwd_apply(_dt(x), _ptr(dout), _ptr(x), _ptr(scale), _ptr(shift), _ptr(mean), _ptr(rstd),
                                          _ptr(c1), _ptr(c2), _ptr(dx), p, cp, ctx.act, slope, None, int(ctx.train),
                                          _stream()))
        return dx, dgamma, dbeta, None, None, None, None, None


class Downsample(torch.autograd.Function):
    """Fixed-kernel strided depthwise correlation with ReplicationPad2d on fp32 NCHW (utils/downsampler.py:65-71)."""

    @staticmethod
    def forward(ctx, x, kern, factor, pad):
        _need_gpu(x)
        x = x.contiguous().float()
        n, c, h, w = x.shape
        k = kern.shape[0]
        oh, ow = (h + 2 * pad - k) // factor + 1, (w + 2 * pad - k) // factor + 1
        y = torch.empty((n, c, oh, ow), dtype=torch.float32, device=x.device)
        check(_lib.lib().dsr_downsample_fwd(_ptr(x), _ptr(kern), _ptr(y), n * c, h, w, k, factor, pad, _stream()))
        ctx.args = (n, c, h, w, k, factor, pad)
        ctx.save_for_backward(kern)
        return y

    @staticmethod
    def backward(ctx, dy):
        (kern,) = ctx.saved_tensors
        n, c, h, w, k, factor, pad = ctx.args
        dy = dy.contiguous().float()
        dx = torch.empty((n, c, h, w), dtype=torch.float32, device=dy.device)
        check(_lib.lib().dsr_downsample_bwd(_ptr(dy), _ptr(kern), _ptr(dx), n * c, h, w, k, factor, pad, _stream()))
        return dx, None, None, None
