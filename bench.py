#!/usr/bin/env python3
"""bench.py -- HR-pixel throughput of the x4 SRGAN train step on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--workload gan_x4|gen_l1_x4|infer_x8|dip_x2]

A "step" is one pass of the hot path over one synthetic batch already resident in HBM:
  gan_x4    : BASELINE config 3 (the configuration the metric is quoted on) -- full GAN step (train_GAN.py:38-71 recipe:
              D step, G step with VGG19 content + adversarial loss, Adam for both), batch 32 per GPU, LR 128x128 ->
              HR 512x512, bf16 storage / fp32 accumulate.
  gen_l1_x4 : BASELINE config 2 -- generator-only L1 step, batch 16, 32x32 -> 128x128.
  infer_x8  : BASELINE config 5 -- Generator(8).eval(), fp16, 256x256 -> 2048x2048.
  dip_x2    : BASELINE config 1 on the GPU -- one Deep-Image-Prior iteration, HR 128x128.
N > 1: one process per GPU, batch sharded data-parallel (weak scaling: fixed per-GPU batch), gradients averaged with RCCL.
The driver launches the ranks with torch.distributed.run; `python bench.py --gpus N` on its own starts them itself (as a
child process, before this process touches the GPU) and relays rank 0's JSON line.  Rank 0 prints ONE JSON line.

Extra legs (rank 0, N = 1 only; each can be switched off):
  roofline      : three more steps, not timed for throughput, with EVERY kernel launch bracketed by HIP events on its launch
                  stream.  Convolution launches are grouped by the kernel the dispatcher selects; for the family with the
                  largest total time: achieved = sum of algorithmic FLOPs (2*N*OH*OW*Cout*KH*KW*Cin) / sum of durations,
                  against the dense bf16 MFMA peak (2.5 PFLOP/s).  Also: whole-step algorithmic FLOP/s over peak
                  (`step_frac`), GPU-busy milliseconds (sum of all kernel brackets) and the launch-gap share of the step.
  cpu_baseline  : the CPU oracle (oracle/, fp32 restatement pinned to the reference) timed on the host cores on a bounded
                  sample of the same workload: warm-up + >= 3 timed steps.
  psnr_delta_db : |PSNR(HIP output, HR) - PSNR(oracle fp32 output, HR)| after K identical steps from identical closed-form
                  weights and inputs, on the reduced configuration named in `psnr_delta.config`.
  other_workloads : (default workload only) the other three configurations, 3 warm-up + 20 timed steps each in this same
                  process: {name: {ms_per_step, value, step_frac}} -- so that every throughput figure DESIGN.md quotes is
                  timed by whoever runs this file, not only by its author.
  fed_from_patch_bank : (default workload only) the same step with its LR / HR batch cut fresh from an HBM-resident uint8
                  image bank before every step (dataset.PatchBank: the real-data path, dataset.py:121-159) instead of one
                  resident synthetic batch -- `value` above keeps the resident-batch definition, this reports the other.
"""
import argparse
import importlib
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PKG = "deep-super-resolution_amd"
MFMA_BF16_PEAK_TFLOPS = 2500.0      # MI355X dense bf16/fp16 (MI355X_MICROARCH.md, chip-level parameters)


def P(sub):
    return importlib.import_module(PKG + "." + sub)


WORKLOADS = {
    # name: batch per GPU, LR size, factor, description, algorithmic GFLOP per step (minimal rows)
    "gan_x4": dict(batch=32, lr=128, factor=4, gflop=32 * 684.87,
                   desc="full GAN step x4 (G + D + VGG19 perceptual), batch 32/GPU, 128x128->512x512"),
    "gen_l1_x4": dict(batch=16, lr=32, factor=4, gflop=217.55,
                      desc="generator-only L1 step x4, batch 16/GPU, 32x32->128x128"),
    "infer_x8": dict(batch=1, lr=256, factor=8, gflop=697.82,
                     desc="generator x8 eval forward, fp16, 256x256->2048x2048 (config 5)"),
    "dip_x2": dict(batch=1, lr=64, factor=2, gflop=28.8,
                   desc="Deep-Image-Prior iteration x2, HR 128x128, fp16 storage (config 1 on the GPU)"),
}


def conv_flops(d):
    n, h, w, cin, cout, kh, kw, stride, pad = d
    oh, ow = (h + 2 * pad - kh) // stride + 1, (w + 2 * pad - kw) // stride + 1
    return 2.0 * n * oh * ow * cout * kh * kw * cin


def pmc_traffic(workload, kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes of this same command
    (profiles/*_traffic_<workload>.json, written by tools/pmc_traffic.py); None when no such profile exists.
    PMC collection needs the profiler around the process, so it cannot be taken live inside the timed run."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_traffic_{workload}.json")))
    if not files:
        return None
    fam = json.load(open(files[-1])).get("families", {}).get(kernel)
    return fam["bytes_per_launch"] if fam else None


def graphs_on():
    return os.environ.get("DSR_HIP_GRAPH", "1") != "0"


def build_step(workload, dev, world):
    """Returns (step callable, HR pixels per rank and step).  `step.eager` is the un-captured form (roofline leg),
    `step.syncs` the GradSync objects (collective-wait timing), `step.modules` the trained networks."""
    cfg = WORKLOADS[workload]
    Gm, optim, steps = P("models.GAN.generator"), P("optim"), P("steps")
    torch.manual_seed(0)                      # module-default init under seed 0

    def finish(fn, px, graph):
        step = steps.GraphedStep(fn) if graph else fn
        if graph:
            step = _Callable(step)
        step.eager = fn
        return step, px

    if workload == "infer_x8":
        infer = P("infer")
        gen = Gm.Generator(8, 16).to(dev)
        img = torch.rand(1, 3, cfg["lr"], cfg["lr"], generator=torch.Generator().manual_seed(1)).to(dev)
        # ~90 small launches per image: replay them from a HIP graph
        return finish(lambda: infer.super_resolve(gen, img), (cfg["lr"] * 8) ** 2, graphs_on())
    if workload == "dip_x2":
        M, Dn = P("models.DIP"), P("utils.downsampler")
        hr_sz = cfg["lr"] * 2
        net = M.get_net(32, "skip", "reflection", upsample_mode="bilinear").to(dev).train()
        down = Dn.Downsampler(3, 2, "lanczos2", phase=0.5, preserve_size=True).to(dev)
        gcpu = torch.Generator().manual_seed(1)
        hr_img = torch.rand(1, 3, hr_sz, hr_sz, generator=gcpu).to(dev)
        with torch.no_grad():
            lr_img = down(hr_img)
        z = (torch.rand(1, 32, hr_sz, hr_sz, generator=gcpu) * 0.1).to(dev)
        run = steps.DipRunner(net, down, z, lr_img, 0.01, 0.05)
        return finish(lambda: run.step()[0], hr_sz * hr_sz, graphs_on())
    gen = Gm.Generator(cfg["factor"], 16).to(dev).train()
    n, s, f = cfg["batch"], cfg["lr"], cfg["factor"]
    g = torch.Generator(device="cpu").manual_seed(1 + (dist.get_rank() if world > 1 else 0))
    lr = torch.rand(n, 3, s, s, generator=g).to(dev)                     # LR ~ U(0,1)
    hr = (torch.rand(n, 3, s * f, s * f, generator=g) * 2 - 1).to(dev)   # HR ~ U(-1,1)
    D = P("dist")
    opt_g = optim.FusedAdam(gen.parameters(), lr=1e-4)
    sync_g = D.GradSync(gen.parameters()).attach()
    F = P("functional")
    if workload == "gen_l1_x4":
        D.broadcast_module(gen)

        def step():
            fake = gen(lr)
            loss = F.l1_loss(fake, hr)
            opt_g.zero_grad()
            with F.batched_wgrad():
                loss.backward()
            sync_g()
            opt_g.step()
            return loss
        # launch-bound workload (~500 launches per step): capture the whole step once, replay it per step
        out, px = finish(step, n * (s * f) ** 2, world == 1 and graphs_on())
        out.syncs, out.modules = [sync_g], [gen]
        return out, px
    Dm, GANu = P("models.GAN.discriminator"), P("utils.GAN")
    disc = Dm.Discriminator((s * f, s * f)).to(dev).train()
    perc = GANu.PerceptualLoss().to(dev)
    D.broadcast_module(gen)
    D.broadcast_module(disc)
    # dense1's 2.1 GB gradient stays in factored form and Adam is applied inside its contraction (same arithmetic bit for bit,
    # tests/test_gpu_models.py::test_fused_dense_adam_equals_separate_launches); DSR_FUSE_DENSE_ADAM=0: two launches
    opt_d = optim.FusedAdam(disc.parameters(), lr=1e-4, fuse_dense_head=os.environ.get("DSR_FUSE_DENSE_ADAM", "1") != "0")
    sync_d = D.GradSync(disc.parameters()).attach()
    overlap = os.environ.get("DSR_GAN_OVERLAP", "1") != "0"

    def step():
        return steps.gan_step(gen, disc, perc, opt_g, opt_d, lr, hr, sync_g, sync_d, overlap=overlap)[1]
    # the whole two-stream step replays from ONE HIP graph (DSR_GAN_GRAPH=0: eager); collectives are not captured
    out, px = finish(step, n * (s * f) ** 2, world == 1 and graphs_on() and os.environ.get("DSR_GAN_GRAPH", "1") != "0"
                     and os.environ.get("DSR_DIST_FORCE", "0") != "1")
    out.modules, out.syncs = [gen, disc], [sync_g, sync_d]
    out.inputs = (lr, hr)
    # the roofline leg times kernels one at a time: on the single-stream form of the same step (identical launches,
    # identical arithmetic) a launch's HIP-event bracket is not stretched by kernels of the other stream
    out.eager = lambda: steps.gan_step(gen, disc, perc, opt_g, opt_d, lr, hr, sync_g, sync_d, overlap=False)[1]
    return out, px


class _Callable:
    """A graph-replayed step that can carry attributes (eager form, modules, syncs)."""

    def __init__(self, fn):
        self._fn = fn

    def __call__(self):
        return self._fn()


def host_cores():
    """Cores this process may actually use: affinity mask, clipped by the cgroup CPU quota (the GPU box gives one
    GPU's share of a large host; os.cpu_count() would oversubscribe it by an order of magnitude)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 64))


def random_state(shapes, seed):
    """Reference-format state_dict with plausible random values (throughput legs only: values do not matter there, and
    hashing 541 M closed-form values would cost more than the measurement)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, shp in shapes.items():
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.zeros(shp, dtype=torch.int64)
        elif k.endswith("running_var"):
            sd[k] = 1.0 + 0.3 * (torch.rand(shp, generator=g) * 2 - 1)
        elif k.endswith("running_mean") or k.endswith("bias"):
            sd[k] = 0.1 * (torch.rand(shp, generator=g) * 2 - 1)
        elif len(shp) >= 2:
            fan_in = 1
            for d in shp[1:]:
                fan_in *= d
            sd[k] = torch.empty(shp).uniform_(-1.0, 1.0, generator=g).mul_((3.0 / fan_in) ** 0.5)
        elif tuple(shp) == (1,):
            sd[k] = torch.full(shp, 0.25)
        else:
            sd[k] = 1.0 + 0.2 * (torch.rand(shp, generator=g) * 2 - 1)
    return sd


def cpu_baseline(workload):
    """The oracle (CPU fp32 restatement of the reference) on a bounded sample of the same workload: one warm-up step,
    then >= 3 timed ones (20 DIP iterations)."""
    from oracle import dip, downsampler, gan, recipes, vgg
    cores = host_cores()
    torch.set_num_threads(cores)
    cfg = WORKLOADS[workload]
    f = cfg["factor"]
    torch.manual_seed(0)
    base = {"unit": "HR Mpixels/s", "cores": cores, "kind": "port"}
    if workload == "infer_x8":
        s, reps = 128, 3                         # quarter-size image: the eval forward is linear in pixels
        gsd = random_state(gan.generator_shapes(8, 16), 1)
        x = torch.rand(1, 3, s, s)
        with torch.no_grad():
            gan.generator_forward(gsd, x, False)
            t0 = time.perf_counter()
            for _ in range(reps):
                gan.generator_forward(gsd, x, False)
        dt = (time.perf_counter() - t0) / reps
        return dict(base, value=(s * 8) ** 2 / dt / 1e6, seconds_per_step=dt,
                    sample=f"{reps} eval forwards of a {s}x{s} LR image after 1 warm-up (workload: 256x256; cost is linear in pixels)")
    if workload == "dip_x2":
        hr_sz = cfg["lr"] * 2
        dcfg = dip.SkipConfig(input_depth=32)
        st = recipes.DipState(random_state(dip.skip_shapes(dcfg), 1), dcfg, torch.rand(1, 32, hr_sz, hr_sz) * 0.1,
                              factor=2, lr=0.01, reg_noise_std=0.05)
        lr_img = downsampler.downsampler_forward(torch.rand(1, 3, hr_sz, hr_sz), 2, "lanczos2", phase=0.5, preserve_size=True)
        recipes.dip_step(st, lr_img)
        t0 = time.perf_counter()
        for _ in range(20):
            recipes.dip_step(st, lr_img)
        dt = (time.perf_counter() - t0) / 20
        return dict(base, value=hr_sz * hr_sz / dt / 1e6, seconds_per_step=dt, sample="20 iterations after 1 warm-up")
    gsd = random_state(gan.generator_shapes(f, 16), 1)
    if workload == "gen_l1_x4":
        n, s, reps = cfg["batch"], cfg["lr"], 3
        st = recipes.GenOnlyState(gsd, lr=1e-4)
        lr, hr = torch.rand(n, 3, s, s), torch.rand(n, 3, s * f, s * f) * 2 - 1
        recipes.gen_l1_step(st, lr, hr)
        t0 = time.perf_counter()
        for _ in range(reps):
            recipes.gen_l1_step(st, lr, hr)
        dt = (time.perf_counter() - t0) / reps
        return dict(base, value=n * (s * f) ** 2 / dt / 1e6, seconds_per_step=dt,
                    sample=f"{reps} full steps (batch {n}, {s}x{s}->{s*f}x{s*f}) after 1 warm-up")
    # config 3 at batch 2 (the batch-32 step needs ~2.5 min per step on these cores).  One part of the step does NOT
    # scale with the batch: Adam over D's 541.6 M parameters (dense1 = 1024 x 524,288); it is timed separately so that
    # the per-sample cost and the batch-32 extrapolation are stated, not assumed.
    n, s, reps = 2, cfg["lr"], 3
    dsd = random_state(gan.discriminator_shapes((s * f, s * f)), 2)
    vsd = random_state(vgg.vgg_shapes(), 3)
    st = recipes.GanState(gsd, dsd, vsd, lr=1e-4)
    fixed = [0.0]
    inner = st.opt_d.step

    def timed_step(*a, **k):
        t = time.perf_counter()
        r = inner(*a, **k)
        fixed[0] += time.perf_counter() - t
        return r
    st.opt_d.step = timed_step
    lr, hr = torch.rand(n, 3, s, s), torch.rand(n, 3, s * f, s * f) * 2 - 1
    recipes.gan_step(st, lr, hr)
    fixed[0] = 0.0
    t0 = time.perf_counter()
    for _ in range(reps):
        recipes.gan_step(st, lr, hr)
    dt = (time.perf_counter() - t0) / reps
    t_fixed = fixed[0] / reps
    per_sample = (dt - t_fixed) / n
    full = t_fixed + cfg["batch"] * per_sample
    px1 = (s * f) ** 2
    return dict(base, value=cfg["batch"] * px1 / full / 1e6, seconds_per_step=dt,
                sample=f"{reps} steps at batch {n} of the batch-{cfg['batch']} workload ({s}x{s}->{s*f}x{s*f}) after 1 warm-up; "
                       f"value = batch-{cfg['batch']} rate extrapolated as fixed + {cfg['batch']} x per-sample",
                measured_value_batch2=n * px1 / dt / 1e6, batch_independent_seconds=t_fixed,
                batch_independent_part="Adam over the discriminator (541.6 M parameters, dense1 = 1024 x 524288)",
                per_sample_seconds=per_sample, extrapolated_seconds_per_step_batch32=full)


def psnr_delta(workload, dev):
    """HIP path vs CPU oracle after K identical steps from identical closed-form weights / inputs (oracle/filler.py), on a
    reduced configuration the oracle finishes in seconds.  PSNR = 10 log10(range^2 / MSE(output, HR target))."""
    from oracle import dip, downsampler, filler, gan, losses, recipes
    Gm, optim, steps = P("models.GAN.generator"), P("optim"), P("steps")
    torch.set_num_threads(host_cores())
    out = {}
    if workload == "gan_x4":
        Dm, GANu = P("models.GAN.discriminator"), P("utils.GAN")
        n, s, f, k = 2, 64, 4, 2
        gsd = filler.fill_state_dict(gan.template(gan.generator_shapes(f, 16)))
        dsd = filler.fill_state_dict(gan.template(gan.discriminator_shapes((s * f, s * f))))
        g, d = Gm.Generator(f, 16), Dm.Discriminator((s * f, s * f))
        g.load_state_dict(gsd), d.load_state_dict(dsd)
        g.to(dev).train(), d.to(dev).train()
        perc = GANu.PerceptualLoss().to(dev)
        vsd = {kk[len("vgg_loss.net.0."):]: v.detach().cpu().clone() for kk, v in perc.state_dict().items()}
        og, od = optim.FusedAdam(g.parameters(), lr=1e-4), optim.FusedAdam(d.parameters(), lr=1e-4)
        st = recipes.GanState(gsd, dsd, vsd, lr=1e-4)
        lr = filler.tensor("bench:lr", (n, 3, s, s), 0.5, 0.5)
        hr = filler.tensor("bench:hr", (n, 3, s * f, s * f))
        deltas, cross = [], []
        for _ in range(k):
            _, _, fake = steps.gan_step(g, d, perc, og, od, lr.to(dev), hr.to(dev))
            _, _, rfake = recipes.gan_step(st, lr, hr)
            deltas.append(abs(losses.psnr(fake.cpu(), hr) - losses.psnr(rfake, hr)))
            cross.append(losses.psnr(fake.cpu(), rfake))
        out["config"] = (f"full GAN step x4, batch {n}, {s}x{s}->{s*f}x{s*f}, Generator(4,16), Discriminator(({s*f},{s*f})), "
                         f"VGG 256/224 stand-in trunk, Adam 1e-4, {k} steps, bf16 storage vs fp32 oracle")
    elif workload == "gen_l1_x4":
        n, s, f, k = 16, 32, 4, 3
        gsd = filler.fill_state_dict(gan.template(gan.generator_shapes(f, 16)))
        g = Gm.Generator(f, 16)
        g.load_state_dict(gsd)
        g.to(dev).train()
        opt = optim.FusedAdam(g.parameters(), lr=1e-4)
        st = recipes.GenOnlyState(gsd, lr=1e-4)
        lr = filler.tensor("bench:lr", (n, 3, s, s), 0.5, 0.5)
        hr = filler.tensor("bench:hr", (n, 3, s * f, s * f))
        deltas, cross = [], []
        for _ in range(k):
            _, fake = steps.gen_l1_step(g, opt, lr.to(dev), hr.to(dev))
            _, rfake = recipes.gen_l1_step(st, lr, hr)
            deltas.append(abs(losses.psnr(fake.cpu(), hr) - losses.psnr(rfake, hr)))
            cross.append(losses.psnr(fake.cpu(), rfake))
        out["config"] = f"the workload itself (batch {n}, {s}x{s}->{s*f}x{s*f}), {k} Adam steps, bf16 storage vs fp32 oracle"
    elif workload == "infer_x8":
        infer = P("infer")
        s = 128
        gsd = filler.fill_state_dict(gan.template(gan.generator_shapes(8, 16)))
        g = Gm.Generator(8, 16)
        g.load_state_dict(gsd)
        g.to(dev)
        x = filler.tensor("bench:lr", (1, 3, s, s), 0.5, 0.5)
        hr = filler.tensor("bench:hr", (1, 3, 8 * s, 8 * s))
        y = infer.super_resolve(g, x.to(dev)).cpu()
        with torch.no_grad():
            ry = gan.generator_forward(gsd, x, False)
        deltas, cross = [abs(losses.psnr(y, hr) - losses.psnr(ry, hr))], [losses.psnr(y, ry)]
        out["config"] = f"Generator(8,16).eval() on a {s}x{s} image, fp16 storage vs fp32 oracle"
    else:
        M, Dn = P("models.DIP"), P("utils.downsampler")
        k, hs = 5, 128
        cfg = dip.SkipConfig(input_depth=32)
        sd = filler.fill_state_dict(gan.template(dip.skip_shapes(cfg)))
        net = M.get_net(32, "skip", "reflection", upsample_mode="bilinear")
        net.load_state_dict(sd)
        net.to(dev).train()
        down = Dn.Downsampler(3, 2, "lanczos2", phase=0.5, preserve_size=True).to(dev)
        hr = filler.tensor("bench:hr", (1, 3, hs, hs), 0.5, 0.5)
        lr_img = downsampler.downsampler_forward(hr, 2, "lanczos2", phase=0.5, preserve_size=True)
        zin = filler.tensor("bench:z", (1, 32, hs, hs), 0.05, 0.05)
        run = steps.DipRunner(net, down, zin.to(dev), lr_img.to(dev), 0.01, 0.05)
        st = recipes.DipState(sd, cfg, zin.clone(), factor=2, lr=0.01, reg_noise_std=0.05)
        deltas, cross = [], []
        for it in range(k):
            noise = filler.tensor(f"bench:noise{it}", (1, 32, hs, hs), 1.7)
            _, o = run.step(noise.to(dev))
            _, ro = recipes.dip_step(st, lr_img, noise)
            deltas.append(abs(losses.psnr(o.cpu(), hr, 1.0) - losses.psnr(ro, hr, 1.0)))
            cross.append(losses.psnr(o.cpu(), ro, 1.0))
        out["config"] = f"the workload itself (HR {hs}x{hs}), {k} iterations with injected noise, fp16 storage vs fp32 oracle"
    torch.cuda.synchronize()
    out.update(value=max(deltas), per_step=deltas, psnr_hip_vs_oracle_db=cross, bar_db=0.02)
    return out


def roofline(step, workload, ms_per_step):
    """An eager step with every launch bracketed by HIP events on its launch stream (see module docstring)."""
    F, L = P("functional"), P("_lib")
    fn = getattr(step, "eager", step)
    fn()                                   # (eager warm-up: packed-weight caches, allocator)
    torch.cuda.synchronize()
    # three bracketed steps: per kernel family the median of its three totals is reported (one step's two grouped
    # weight-gradient launches alone read 1.64 .. 1.81 ms from run to run on this power-managed chip), busy time likewise
    runs = []
    for _ in range(3):
        F.KERNEL_LOG, L.LAUNCH_LOG = [], []
        # DSR_BENCH_SPIN=1 (probe): the host issues the step while the GPU spins, so that every bracket opens when the previous
        # launch ends with its own launch already queued (no host time inside a bracket, the chip continuously loaded as in
        # the replayed step).  Measured on one box against the rocprofv3 summary of the same command: the grouped weight
        # gradient reads the same either way (3.22-3.34 ms per step, rocprof 3.34), the 14 launches of the 256x256 tile read
        # 3.38-3.40 with the spin and 3.20-3.28 without (rocprof 3.12): off by default, the brackets closest to the trace.
        if os.environ.get("DSR_BENCH_SPIN", "0") == "1":
            torch.cuda._sleep(int(0.10 * 2.0e9))
        fn()
        torch.cuda.synchronize()
        runs.append((sum(e0.elapsed_time(e1) for _, e0, e1 in L.LAUNCH_LOG), F.KERNEL_LOG, L.LAUNCH_LOG))
    F.KERNEL_LOG = L.LAUNCH_LOG = None
    _, conv_log, all_log = sorted(runs, key=lambda r: r[0])[1]
    med = lambda v: sorted(v)[len(v) // 2]
    # An event pair with nothing between its two records still reads a few microseconds (the markers themselves): measured
    # here and taken off every kernel bracket, so that the per-kernel averages agree with a rocprofv3 --kernel-trace summary
    # of the same command (profiles/r02_gan_x4_serial_kernel_stats.csv) instead of sitting ~8 % above it.
    empty = []
    for _ in range(64):
        a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a0.record()
        a1.record()
        empty.append((a0, a1))
    torch.cuda.synchronize()
    ev_ms = sorted(x.elapsed_time(y) for x, y in empty)[len(empty) // 2]
    def families(log):
        fam = {}
        for kind, d, e0, e1, k in log:
            t, fl, cnt = fam.get(k, (0.0, 0.0, 0))
            # one C-ABI call = one kernel launch, except a strided dgrad on the gather kernel (stride^2 parity classes)
            nl = d[7] * d[7] if (kind == "dgrad" and k.startswith("conv_gemm")) else 1
            flops = sum(conv_flops(q) for q in d) if kind == "wgrad_batch" else conv_flops(d)   # (a grouped launch: many layers)
            fam[k] = (t + max(e0.elapsed_time(e1) - ev_ms, 0.0) * 1e-3, fl + flops, cnt + nl)
        return fam
    fams = [families(r[1]) for r in runs]
    fam = {k: (med([f[k][0] for f in fams if k in f]), v[1], v[2]) for k, v in families(conv_log).items()}
    if not fam:
        return None
    busy = med([r[0] for r in runs])
    by_entry = {}
    for name, e0, e1 in all_log:
        by_entry[name] = by_entry.get(name, 0.0) + e0.elapsed_time(e1)
    top = max(fam, key=lambda k: fam[k][0])
    t, fl, cnt = fam[top]
    ach = fl / t / 1e12
    step_tf = WORKLOADS[workload]["gflop"] / ms_per_step            # GFLOP / ms = TFLOP/s
    return {"bound": "mfma", "kernel": top, "achieved": ach, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": ach / MFMA_BF16_PEAK_TFLOPS, "traffic": pmc_traffic(workload, top),
            "launches": cnt, "avg_launch_ms": t / cnt * 1e3, "event_pair_overhead_ms": ev_ms,
            "measured_on": "three eager single-stream runs of the step, per kernel family the median of its three totals "
                           "(per-kernel HIP-event brackets are not stretched by a concurrent stream); the timed steps "
                           "replay the same launches",
            "step_algorithmic_tflops": step_tf, "step_frac": step_tf / MFMA_BF16_PEAK_TFLOPS,
            "gpu_busy_ms": busy, "launches_total": len(all_log),
            "launch_gap_share": max(0.0, 1.0 - busy / ms_per_step) if busy < ms_per_step else 0.0,
            "busy_ms_by_entry_point": {k: round(v, 3) for k, v in sorted(by_entry.items(), key=lambda kv: -kv[1])[:12]},
            "families": {k: {"seconds": v[0], "tflops": v[1] / v[0] / 1e12, "launches": v[2]} for k, v in fam.items()}}


def time_workload(name, dev, steps=20, warmup=3):
    """ms per step of one of the other workloads in this process (same timing rule: sync on both sides of `steps` steps)."""
    step, px = build_step(name, dev, 1)
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    step_tf = WORKLOADS[name]["gflop"] / ms
    return {"ms_per_step": ms, "value": px / ms / 1e3, "unit": "HR Mpixels/s", "steps": steps, "warmup": warmup,
            "step_frac": step_tf / MFMA_BF16_PEAK_TFLOPS, "hip_graph": bool(isinstance(step, _Callable)),
            "workload": WORKLOADS[name]["desc"]}


def time_fed_from_patch_bank(step, dev, steps, px):
    """The config-3 step with a FRESH batch per step: 32 LR / HR patch pairs cut from an HBM-resident uint8 image bank
    (dataset.PatchBank: host-side draws of image and position + two byte-kernel launches) and copied into the step's input
    tensors, then the same (graph-replayed) step.  100 pre-shrunk pairs of 170x255 / 680x1020 pixels: a DIV2K x8 file halved
    twice, as the reference's loader shrinks them (dataset.py:24-49)."""
    import numpy as np
    DS, Dg = P("dataset"), P("utils.degradation")
    lr, hr = step.inputs
    rng = np.random.RandomState(0)
    pairs = []
    for _ in range(100):
        img = torch.from_numpy(rng.randint(0, 256, (170, 255, 3), dtype=np.uint8)).to(dev)
        pairs.append((img, Dg.resize(img, 255 * 4, 170 * 4)))
    bank = DS.PatchBank(pairs, 4, (lr.shape[3], lr.shape[2]), reference_scaling=False, rng=np.random.RandomState(1))

    def fed():
        a, b = bank.sample(lr.shape[0])
        lr.copy_(a)
        hr.copy_(b)
        return step()
    fed()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fed()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    return {"ms_per_step": ms, "value": px / ms / 1e3, "unit": "HR Mpixels/s", "steps": steps,
            "feed": "PatchBank.sample(32) from 100 HBM-resident uint8 pairs + copy into the step's inputs, every step"}


def spawn_ranks(a):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a CHILD process tree (this process has not
    touched the GPU and never will) and hand back rank 0's output and exit code."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=dict(os.environ, MASTER_ADDR="127.0.0.1")).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="gan_x4", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-psnr", action="store_true")
    ap.add_argument("--no-other-workloads", action="store_true")
    a = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        raise SystemExit(spawn_ranks(a))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world and os.environ.get("DSR_DIST_FORCE", "0") != "1":
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # DSR_DIST_REHEARSAL=1 (development aid, one-GPU box): every rank uses cuda:0 and the ranks talk over gloo, so the
    # N > 1 code path (broadcast, gradient hooks, bucketed all-reduce, two-stream step) runs end to end without RCCL
    rehearsal = os.environ.get("DSR_DIST_REHEARSAL", "0") == "1"
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    force = os.environ.get("DSR_DIST_FORCE", "0") == "1"     # development aid: the RCCL path on a world of one rank
    if world > 1 or force:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            # no device_id=: binding the group to the device at init (eager communicator creation) made every step
            # 3.3 ms slower on this stack, collectives or not (measured, world of one rank); the lazily created
            # communicator does not.  The device is already selected by torch.cuda.set_device(local) above.
            dist.init_process_group("nccl")
    P("_lib").lib()

    step, px_per_rank = build_step(a.workload, dev, world)

    def fence():
        if world > 1 or force:
            # (device_ids: the group is not bound to a device at init, so name the one this rank's barrier runs on)
            dist.barrier(device_ids=[local]) if dist.get_backend() == "nccl" else dist.barrier()
        torch.cuda.synchronize()

    def note(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    note(f"workload {a.workload}, world {world}: warm-up")
    for _ in range(a.warmup):
        step()
    fence()
    syncs = getattr(step, "syncs", [])
    for sy in syncs:
        sy.timing = world > 1 or force
    note("timed region")
    # two (shader-clock cycles, 100 MHz ticks) samples bracket the timed steps on the stream: the clock the chip actually held
    clk = torch.zeros(32, dtype=torch.int64, device=dev)
    L = P("_lib")
    st_ptr = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    t0 = time.perf_counter()
    L.check(L.lib().dsr_clock_sample(ctypes.c_void_p(clk.data_ptr()), st_ptr))
    for _ in range(a.steps):
        step()
    L.check(L.lib().dsr_clock_sample(ctypes.c_void_p(clk.data_ptr() + 128), st_ptr))
    t_issue = time.perf_counter() - t0          # host time to ISSUE the steps (GPU still running): launch-bound if ~ dt
    fence()
    dt = time.perf_counter() - t0
    c = clk.cpu().tolist()
    per_xcd = [(c[16 + 2 * x] - c[2 * x]) / (c[17 + 2 * x] - c[1 + 2 * x]) * 0.1            # cycles per 10 ns tick
               for x in range(8) if c[1 + 2 * x] and c[17 + 2 * x] > c[1 + 2 * x]]
    # (the cycle counters of different CUs are not synchronised: the two samples of an XCD may come from different CUs, a
    #  constant offset that only washes out over a long region -- below 0.2 s the figure is not reported)
    shader_ghz = sum(per_xcd) / len(per_xcd) if (per_xcd and dt >= 0.2) else None
    wait_ms = sum(sy.pop_wait_ms() for sy in syncs) / a.steps
    for sy in syncs:
        sy.timing = False
    if world > 1 or force:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # exposed (not overlapped) collective time per rank: how long the compute streams sat behind communication
        w = torch.tensor([wait_ms], device=dev, dtype=torch.float64)
        ws = [torch.zeros_like(w) for _ in range(world)]
        if dist.get_backend() == "gloo":
            w, ws = w.cpu(), [x.cpu() for x in ws]
        dist.all_gather(ws, w)
        wait_all = [float(x.item()) for x in ws]
        print(f"[bench] rank {rank}: collective wait {wait_ms:.3f} ms/step (stream-side, exposed)", file=sys.stderr, flush=True)
    ms = dt / a.steps * 1e3
    value = px_per_rank * world * a.steps / dt / 1e6
    if (rehearsal or force or os.environ.get("DSR_BENCH_CHECKSUM", "0") == "1") and hasattr(step, "modules"):
        # every rank must hold bit-identical parameters after the same averaged updates
        for m in step.modules:
            cs = torch.stack([p.detach().double().abs().sum() for p in m.parameters()]).sum().reshape(1)
            lo, hi = cs.clone(), cs.clone()
            if world > 1 or force:
                if dist.get_backend() == "gloo":
                    lo, hi = lo.cpu(), hi.cpu()
                dist.all_reduce(lo, op=dist.ReduceOp.MIN)
                dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            assert torch.isfinite(cs).all() and lo.item() == hi.item(), ("ranks diverged", lo.item(), hi.item())
            note(f"rehearsal: checksum {type(m).__name__} {cs.item():.12e}")
        note("rehearsal: parameters identical on all ranks")

    metric = {"infer_x8": "HR Mpixels/sec x8 generator inference", "dip_x2": "HR Mpixels/sec DIP iteration"}.get(
        a.workload, "HR Mpixels/sec x4 GAN train step; PSNR delta vs reference")
    out = {"metric": metric, "value": value, "unit": "HR Mpixels/s", "n_gpus": world,
           "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f16" if a.workload in ("infer_x8", "dip_x2") else "bf16", "data": "synthetic",
           "config": {"workload": a.workload + ": " + WORKLOADS[a.workload]["desc"],
                      "global_batch": WORKLOADS[a.workload]["batch"] * world, "parallelism": f"dp{world}"},
           "host_issue_ms_per_step": t_issue / a.steps * 1e3,
           "hip_graph": bool(isinstance(step, _Callable)),
           # average shader clock over the timed steps (s_memtime / s_memrealtime): the chip lowers it under load, so a
           # fraction of the 2.5 PFLOP/s nominal peak (2.4 GHz) is really measured against peak * clock / 2.4
           "avg_shader_clock_ghz": None if shader_ghz is None else round(shader_ghz, 3)}
    if world > 1 or force:
        out["collective_wait_ms_per_step"] = wait_all

    note(f"{ms:.2f} ms/step (host issue {t_issue / a.steps * 1e3:.2f} ms/step)")
    if rank == 0 and world == 1 and not a.no_roofline:
        note("roofline leg")
        r = roofline(step, a.workload, ms)
        if r:
            out["roofline"] = r
    if rank == 0 and world == 1 and not a.no_psnr:
        note("psnr delta (HIP vs oracle on the reduced configuration)")
        pd = psnr_delta(a.workload, dev)
        out["psnr_delta_db"] = pd["value"]
        out["psnr_delta"] = pd
    if rank == 0 and world == 1 and not force and a.workload == "gan_x4" and not a.no_other_workloads:
        note("the same step fed from a PatchBank")
        out["fed_from_patch_bank"] = time_fed_from_patch_bank(step, dev, a.steps, px_per_rank)
        out["other_workloads"] = {}
        for name in ("gen_l1_x4", "infer_x8", "dip_x2"):
            note(f"other workload {name}")
            out["other_workloads"][name] = time_workload(name, dev)
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        note("cpu baseline (oracle on host cores)")
        out["cpu_baseline"] = cpu_baseline(a.workload)
    if rank == 0:
        print(json.dumps(out))
    if world > 1 or force:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
