"""Adam with torch.optim.Adam's default semantics (train_GAN.py:35-36, utils/DIP.py:34) as one fused
HIP kernel per tensor; the step counter lives on the device so a whole train step can be captured
in a HIP graph."""
import ctypes as C

import torch

from . import _lib
from .functional import _ptr, _stream, bump, check, mark_shadow_current, repack_cached, shadow_for_update


class FusedAdam:
    MULTI_MAX = 1 << 20      # tensors up to this many elements go through the multi-tensor launch

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, grad_scale=1.0, fuse_dense_head=False):
        self.params = [p for p in params]
        if not self.params:
            raise ValueError("optimizer got an empty parameter list")   # torch.optim raises the same
        # fuse_dense_head: the dense head's big matrix (discriminator.py:54, marked `_dsr_dense_head`) gets its gradient as
        # two 16-bit factors (functional.GradFactors) instead of a `.grad` tensor, and step() applies Adam inside the
        # weight-gradient contraction (dsr_linear_wgrad_adam).  Same arithmetic, bit for bit; `.grad` of that one tensor
        # stays None, which is why it is opt-in (steps / bench turn it on, nothing else reads that gradient).
        self.fuse_dense_head = bool(fuse_dense_head)
        if self.fuse_dense_head:
            for p in self.params:
                if getattr(p, "_dsr_dense_head", False):
                    p._dsr_defer_wgrad = True
        self.lr, self.betas, self.eps = float(lr), betas, float(eps)
        self.grad_scale = float(grad_scale)     # gradients are multiplied by this first (1/S for a loss scale S)
        dev = self.params[0].device
        self.m = [torch.zeros_like(p, memory_format=torch.contiguous_format) for p in self.params]
        self.v = [torch.zeros_like(p, memory_format=torch.contiguous_format) for p in self.params]
        self.step_t = torch.zeros(1, dtype=torch.int32, device=dev)

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            if getattr(p, "_dsr_grad_factors", None):
                p._dsr_grad_factors = []
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def step(self):
        lib = _lib.lib()
        st = _stream()
        check(lib.dsr_pw_incr(_ptr(self.step_t), st))
        small, keep = [], []
        for p, m, v in zip(self.params, self.m, self.v):
            pending = getattr(p, "_dsr_grad_factors", None)
            if pending:
                p._dsr_grad_factors = []
                if len(pending) == 1 and p.grad is None and p.is_contiguous():
                    f = pending[0]
                    f.wait()
                    sh = shadow_for_update(p)
                    check(lib.dsr_linear_wgrad_adam(f.dt, _ptr(f.dyt), _ptr(f.xt), f.bp, f.o, f.k, f.ranks, f.scale, _ptr(p),
                                                    _ptr(m), _ptr(v), _ptr(sh), _ptr(self.step_t), self.lr, self.betas[0],
                                                    self.betas[1], self.eps, self.grad_scale, st))
                    bump(p)
                    if sh is not None:
                        mark_shadow_current(p)
                    continue
                # several backward passes since zero_grad (or a .grad from elsewhere): accumulate like autograd would
                for f in pending:
                    g = f.materialize()
                    p.grad = g if p.grad is None else p.grad + g
            if p.grad is None:
                continue
            g = p.grad
            if g.dtype != torch.float32 or not g.is_contiguous():
                g = g.float().contiguous()
                keep.append(g)
            sh = shadow_for_update(p)        # bf16 image kept by DenseHead for this matrix (or None)
            if sh is None and p.numel() <= self.MULTI_MAX and p.is_contiguous():
                small.append((p, g, m, v))
            else:
                check(lib.dsr_pw_adam(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), self.lr, self.betas[0],
                                      self.betas[1], self.eps, _ptr(self.step_t), self.grad_scale, _ptr(sh), st))
            bump(p)
            if sh is not None:               # after bump(): the shadow written by this launch IS the new version
                mark_shadow_current(p)
        if small:                            # every small tensor in one launch per 64 (dsr_pw_adam_multi)
            k = len(small)
            arr = [(C.c_void_p * k)(*[t[i].data_ptr() for t in small]) for i in range(4)]
            ns = (C.c_size_t * k)(*[t[0].numel() for t in small])
            check(lib.dsr_pw_adam_multi(k, arr[0], arr[1], arr[2], arr[3], ns, self.lr, self.betas[0], self.betas[1],
                                        self.eps, _ptr(self.step_t), self.grad_scale, st))
        repack_cached(self.params)           # packed 16-bit conv weight images follow in one launch, not one per layer
