"""CPU, world_size 2, gloo: the data-parallel gradient averaging (deep-super-resolution_amd/dist.py) gives every
rank the mean of the per-rank gradients, for bucketed small tensors and in-place big ones alike, and is a no-op
when torch.distributed is not initialised."""
import importlib
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = "deep-super-resolution_amd"


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    D = importlib.import_module(PKG + ".dist")
    torch.manual_seed(0)
    shapes = [(64, 64, 3, 3), (64,), (1,), (1024, 300), (3, 64, 9, 9)]
    params = [torch.nn.Parameter(torch.zeros(s)) for s in shapes]
    ref = []
    for i, p in enumerate(params):
        per_rank = [torch.full(p.shape, float(r + 1)) * (i + 1) + torch.arange(p.numel()).reshape(p.shape) * 1e-3 * (r + 1)
                    for r in range(world)]
        p.grad = per_rank[rank].clone()
        ref.append(sum(per_rank) / world)
    sync = D.GradSync(params, bucket_bytes=100_000, big_bytes=1_000_000)     # (1024,300) fp32 = 1.2 MB -> "big" path
    sync()
    ok = all(torch.allclose(p.grad, r, rtol=1e-6, atol=1e-6) for p, r in zip(params, ref))
    # hook path: the big tensor's all-reduce starts from autograd's post-accumulate hook, during backward
    sync2 = D.GradSync(params, bucket_bytes=100_000, big_bytes=1_000_000).attach()
    for p in params:
        p.grad = None
    loss = sum(((rank + 1.0) * (i + 1) * p).sum() for i, p in enumerate(params))
    loss.backward()
    early = len(sync2._pending)
    sync2()
    ok = ok and early == 1 and all(torch.allclose(p.grad, torch.full(p.shape, (1 + world) / 2.0 * (i + 1)))
                                   for i, p in enumerate(params))
    # broadcast_module: rank 1 starts from garbage and must end with rank 0's values
    lin = torch.nn.Linear(4, 3)
    if rank == 1:
        with torch.no_grad():
            for t in lin.parameters():
                t.fill_(123.0)
    D.broadcast_module(lin)
    w = lin.weight.detach().clone()
    gathered = [torch.zeros_like(w) for _ in range(world)]
    dist.all_gather(gathered, w)
    ok = ok and torch.equal(gathered[0], gathered[1])
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_grad_sync_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]


def _factor_worker(rank, world, port, q):
    """The dense head's data-parallel gradient (functional.DenseHead.backward): instead of all-reducing dW1 = dy^T x
    (2.1 GB at 512x512), every rank all-gathers the two rank-local FACTORS and forms (1/R) sum_r dy_r^T x_r itself.
    Here the exchange runs over gloo on CPU-resident factors (the same functional._gather_factors the GPU path calls;
    the MFMA product dsr_linear_wgrad_gathered is replaced by a plain matmul) and must equal the all-reduced mean."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    F = importlib.import_module(PKG + ".functional")
    D = importlib.import_module(PKG + ".dist")
    o, k, bp, b = 24, 200, 32, 5
    g = torch.Generator().manual_seed(100 + rank)
    x = torch.randn(b, k, generator=g)            # this rank's flattened features [batch, K]
    dy = torch.randn(b, o, generator=g)           # this rank's dense1 output gradient [batch, O]
    xt, dyt = torch.zeros(k, bp), torch.zeros(o, bp)        # batch-minor, zero padded: the layout DenseHead exchanges
    xt[:, :b], dyt[:, :b] = x.t(), dy.t()
    xt_all, dyt_all, works = F._gather_factors(xt, dyt, world)
    for wk in works:
        wk.wait()
    ok = tuple(xt_all.shape) == (world, k, bp) and tuple(dyt_all.shape) == (world, o, bp)
    ok = ok and torch.equal(xt_all[rank], xt) and torch.equal(dyt_all[rank], dyt)
    from_factors = sum(dyt_all[r] @ xt_all[r].t() for r in range(world)) / world
    # the plain alternative (DSR_DP_FACTOR_GATHER=0): local gradient, averaged by GradSync's all-reduce
    w1 = torch.nn.Parameter(torch.zeros(o, k))
    w1.grad = dy.t() @ x
    D.GradSync([w1], big_bytes=1)()               # big path: in-place all-reduce
    ok = ok and torch.allclose(from_factors, w1.grad, rtol=1e-5, atol=1e-5)
    # a parameter marked as produced globally (attach(factor_gather=True) on the dense head) is left alone by GradSync
    w2 = torch.nn.Parameter(torch.zeros(o, k))
    w2._dsr_dense_head = True
    sync = D.GradSync([w2], big_bytes=1).attach()
    ok = ok and getattr(w2, "_dsr_grad_global", False) and F.dp_world_for(w2) == world
    w2.grad = torch.full((o, k), float(rank + 1))
    sync()
    ok = ok and torch.equal(w2.grad, torch.full((o, k), float(rank + 1)))
    # optim.FusedAdam(fuse_dense_head=True) keeps dense1's gradient as a factor pair only when nobody will all-reduce it:
    # a globally produced gradient (w2) may stay factored, a plain one (w1: GradSync averages it after backward) may not
    ok = ok and F._awaits_allreduce(w1) and not F._awaits_allreduce(w2)
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_dense_head_factor_gather_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 30100 + (os.getpid() % 500)
    procs = [ctx.Process(target=_factor_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]


def test_grad_sync_noop_single_process():
    sys.path.insert(0, ROOT)
    D = importlib.import_module(PKG + ".dist")
    p = torch.nn.Parameter(torch.zeros(3))
    p.grad = torch.ones(3)
    D.GradSync([p])()
    assert torch.equal(p.grad, torch.ones(3)) and not D.is_dist()
    F = importlib.import_module(PKG + ".functional")
    assert not F._awaits_allreduce(p)            # one process: nothing to average, the dense gradient may stay factored
