"""Minibatch data parallelism for the GAN step: one process per GPU, gradients averaged with RCCL
all-reduce over xGMI (torch.distributed backend "nccl" == RCCL on ROCm); "gloo" on CPU for tests.

The reference has no distributed code at all; BASELINE config 4 asks for exactly this one
strategy.  Semantics: every rank runs train_GAN.py:38-71 on its own shard of the global batch with local
(per-rank) BatchNorm statistics; after each backward the parameter gradients are averaged, so every rank applies
the same Adam update (mean-reduced losses over equal shards average exactly to the global-batch loss).

Bucketing follows the hardware: xGMI is point-to-point, so few large messages beat many small ones, and the one huge
gradient is not exchanged at all: the discriminator's dense1.weight gradient is 2.1 GB at 512x512 (a single xGMI link
between two GPUs would need ~30 ms for its all-reduce), but it is the product of two small rank-local factors
(dy^T: 128 KB, x^T: 67 MB).  functional.DenseHead all-gathers those and forms the rank-averaged gradient itself
(``dsr_linear_wgrad_gathered``); this module marks the tensor and skips it.  Other tensors of at least ``big_bytes``
are reduced in place, each as its own message, from autograd's post-accumulate hook; everything smaller is packed (one
fused multi-tensor copy) into persistent flat buckets of ``bucket_bytes`` whose slices then serve as the .grad tensors, so
nothing is copied back.  RCCL averages inside the collective (ReduceOp.AVG).  Collectives are issued asynchronously and
only waited for right before the optimiser.
"""
import os

import torch
import torch.distributed as dist

# DSR_DIST_FORCE=1 (development aid): treat an initialised world-size-1 process group as data parallel, so that every
# collective of the N > 1 path (broadcast, hook-issued and bucketed all-reduce, factor all-gather) is really issued --
# on a one-GPU box that is the only way to run them through RCCL ("nccl") rather than gloo.
FORCE = os.environ.get("DSR_DIST_FORCE", "0") == "1"


def is_dist():
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or FORCE)


def broadcast_module(module, src=0):
    """Make every rank start from rank `src`'s parameters and buffers (DDP convention)."""
    if not is_dist():
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src)


class GradSync:
    """Average .grad of `params` over the ranks.

    attach()  registers a post-accumulate hook on every tensor of at least ``big_bytes``: its all-reduce is issued
              the moment autograd has finished that gradient, i.e. while the rest of the backward pass is still
              running (the discriminator's dense1.weight gradient -- 99 % of D's gradient bytes -- is the FIRST one
              backward produces, so its ~2 GB exchange hides behind the whole conv backward).
    launch()  (after backward) packs the remaining small gradients into flat buckets and starts their all-reduces.
    wait()    blocks the compute stream until everything has landed and divides by the world size.
    """

    def __init__(self, params, bucket_bytes=64 << 20, big_bytes=32 << 20):
        self.params = [p for p in params if p.requires_grad]
        self.bucket_bytes, self.big_bytes = bucket_bytes, big_bytes
        self._pending = []
        self._early = set()
        self._hooks = []
        self._plan_key, self._buckets = None, []
        # measurement aid (bench.py --gpus N): with `timing` set, wait() brackets the stream-side wait for the
        # collectives with events; pop_wait_ms() returns the accumulated milliseconds the compute stream spent parked
        # behind communication (exposed, i.e. NOT overlapped, collective time) since the last call
        self.timing = False
        self._wait_events = []

    def attach(self, factor_gather=True):
        """``factor_gather``: 2-D parameters of at least ``big_bytes`` (the dense head's K x 1024 matrix) are marked
        ``_dsr_grad_global``: functional.DenseHead then all-gathers the small rank-local factors of that gradient and
        forms the rank-averaged gradient itself, and this object leaves the tensor alone.  Anything else that big is
        all-reduced from its post-accumulate hook."""
        if not is_dist() or self._hooks:
            return self
        factor_gather = factor_gather and os.environ.get("DSR_DP_FACTOR_GATHER", "1") != "0"   # 0: plain all-reduce
        for p in self.params:
            if p.numel() * p.element_size() >= self.big_bytes:
                if factor_gather and p.dim() == 2 and getattr(p, "_dsr_dense_head", False):
                    p._dsr_grad_global = True
                else:
                    self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad_ready))
        return self

    @staticmethod
    def _reduce_op():
        """(op, divide): RCCL averages inside the collective; gloo has no AVG, so sum and divide afterwards."""
        if dist.get_backend() == "nccl":
            return dist.ReduceOp.AVG, False
        return dist.ReduceOp.SUM, True

    def _on_grad_ready(self, p):
        if not is_dist() or p.grad is None:
            return
        op, divide = self._reduce_op()
        work = dist.all_reduce(p.grad, op=op, async_op=True)
        self._pending.append(("big", work, p.grad, None, divide))
        self._early.add(id(p))

    def _plan(self, small):
        """Persistent flat buckets for the small gradients: [(flat, [(param, view of flat shaped like param), ...]), ...].
        Slots start on 256-byte boundaries (the fused Adam reads .grad with 16-byte vector loads)."""
        key = tuple((id(p), p.grad.dtype) for p in small)
        if key == self._plan_key:
            return self._buckets
        groups, cur, size = [], [], 0
        for p in small:
            if cur and (p.grad.dtype != cur[0].grad.dtype or size >= self.bucket_bytes):
                groups.append(cur)
                cur, size = [], 0
            cur.append(p)
            size += p.grad.numel() * p.grad.element_size()
        if cur:
            groups.append(cur)
        self._buckets = []
        for grp in groups:
            align = 256 // grp[0].grad.element_size()
            offs, total = [], 0
            for p in grp:
                offs.append(total)
                total += (p.numel() + align - 1) // align * align
            flat = torch.zeros(total, dtype=grp[0].grad.dtype, device=grp[0].grad.device)
            self._buckets.append((flat, [(p, flat[o:o + p.numel()].view_as(p)) for p, o in zip(grp, offs)]))
        self._plan_key = key
        return self._buckets

    def launch(self):
        """Start averaging every .grad that has not been started by a hook (call right after backward)."""
        if not is_dist():
            return
        op, divide = self._reduce_op()
        small = []
        for p in reversed(self.params):          # reverse registration order ~ order grads became ready
            g = p.grad
            if g is None or id(p) in self._early or getattr(p, "_dsr_grad_global", False):
                continue
            if g.numel() * g.element_size() >= self.big_bytes or not g.is_contiguous():
                self._pending.append(("big", dist.all_reduce(g, op=op, async_op=True), g, None, divide))
            else:
                small.append(p)
        for flat, items in self._plan(small):
            # one fused multi-tensor copy packs the bucket; after the collective the views BECOME the .grad tensors
            # (wait()), so nothing is copied back -- per-tensor copies cost ~140 launches per optimiser and step
            torch._foreach_copy_([v for _, v in items], [p.grad for p, _ in items])
            self._pending.append(("bucket", dist.all_reduce(flat, op=op, async_op=True), flat, items, divide))

    def wait(self):
        """Block the current stream until the averages have landed; afterwards every small parameter's .grad is a view
        of its bucket (valid until the next launch(), like DDP's gradient_as_bucket_view)."""
        world = dist.get_world_size() if self._pending else 1
        timed = self.timing and self._pending and self._pending[0][2].is_cuda
        if timed:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
        for kind, work, buf, items, divide in self._pending:
            work.wait()
            if divide:
                buf.div_(world)
            if kind == "bucket":
                for p, v in items:
                    p.grad = v
        if timed:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            self._wait_events.append((e0, e1))
        self._pending = []
        self._early = set()

    def pop_wait_ms(self):
        """Milliseconds the compute stream waited for collectives in the wait() calls since the last pop (needs a device
        sync by the caller first).  0.0 when nothing was timed."""
        ms = sum(a.elapsed_time(b) for a, b in self._wait_events)
        self._wait_events = []
        return ms

    def __call__(self):
        self.launch()
        self.wait()
