"""Minibatch data parallelism for the GAN step: one process per GPU, gradients averaged with RCCL
all-reduce over xGMI (torch.distributed backend "nccl" == RCCL on ROCm); "gloo" on CPU for tests.

The reference has no distributed code at all; BASELINE config 4 asks for exactly this one
strategy.  Semantics: every rank runs train_GAN.py:38-71 on its own shard of the global batch with local
(per-rank) BatchNorm statistics; after each backward the parameter gradients are averaged, so every rank applies
the same Adam update (mean-reduced losses over equal shards average exactly to the global-batch loss).

Bucketing follows the hardware: xGMI is point-to-point, so few large messages beat many small ones.  Tensors of
at least ``big_bytes`` (the discriminator's dense1.weight gradient is 2.1 GB at 512x512) are reduced in place,
each as its own message, in the order autograd produced them; everything smaller is packed into flat buckets of
``bucket_bytes``.  Collectives are issued asynchronously and only waited for right before the optimiser.
"""
import torch
import torch.distributed as dist


def is_dist():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def broadcast_module(module, src=0):
    """Make every rank start from rank `src`'s parameters and buffers (DDP convention)."""
    if not is_dist():
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src)


class GradSync:
    def __init__(self, params, bucket_bytes=64 << 20, big_bytes=32 << 20):
        self.params = [p for p in params if p.requires_grad]
        self.bucket_bytes, self.big_bytes = bucket_bytes, big_bytes
        self._pending = []

    def launch(self):
        """Start averaging every .grad (call right after backward)."""
        if not is_dist():
            return
        world = dist.get_world_size()
        small, size = [], 0
        for p in reversed(self.params):          # reverse registration order ~ order grads became ready
            g = p.grad
            if g is None:
                continue
            nbytes = g.numel() * g.element_size()
            if nbytes >= self.big_bytes:
                self._pending.append(("big", dist.all_reduce(g, op=dist.ReduceOp.SUM, async_op=True), g, None, world))
                continue
            small.append(g)
            size += nbytes
            if size >= self.bucket_bytes:
                self._flush(small, world)
                small, size = [], 0
        if small:
            self._flush(small, world)

    def _flush(self, grads, world):
        flat = torch.cat([g.reshape(-1) for g in grads])
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)
        self._pending.append(("bucket", work, flat, list(grads), world))

    def wait(self):
        """Block the current stream until the averages have landed in the .grad tensors."""
        for kind, work, buf, grads, world in self._pending:
            work.wait()
            if kind == "big":
                buf.div_(world)
            else:
                buf.div_(world)
                off = 0
                for g in grads:
                    n = g.numel()
                    g.copy_(buf[off:off + n].view_as(g))
                    off += n
        self._pending = []

    def __call__(self):
        self.launch()
        self.wait()
