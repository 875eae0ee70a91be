#!/usr/bin/env python3
"""bench.py -- HR-pixel throughput of the x4 SRGAN train step on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--workload gan_x4|gen_l1_x4]

A "step" is one pass of the hot path over one synthetic batch already resident in HBM:
  gan_x4    : BASELINE config 3 -- full GAN step (train_GAN.py:38-71 recipe: D step, G step with VGG19 content +
              adversarial loss, Adam for both), batch 32 per GPU, LR 128x128 -> HR 512x512, bf16 storage / fp32 accumulate.
  gen_l1_x4 : BASELINE config 2 -- generator-only L1 step, batch 16, 32x32 -> 128x128.
N > 1: one process per GPU (torchrun), batch sharded data-parallel (weak scaling: fixed per-GPU batch), gradients
averaged with RCCL all-reduce.  Rank 0 prints ONE JSON line.

Extra legs (rank 0, N = 1 only):
  roofline     : one more, un-timed-for-throughput step is run with every convolution launch bracketed by HIP events
                 on its own stream; for the kernel family with the largest total time it reports algorithmic
                 FLOP/s = sum of 2*M*Cout*KH*KW*Cin over its launches / sum of their durations, against the dense
                 bf16 MFMA peak (2.5 PFLOP/s).
  cpu_baseline : the CPU oracle (oracle/, a PyTorch fp32 restatement pinned to the reference) timed on the host cores
                 on a bounded sample of the same workload.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PKG = "deep-super-resolution_amd"
MFMA_BF16_PEAK_TFLOPS = 2500.0      # MI355X dense bf16 (MI355X_MICROARCH.md, chip-level parameters)


def P(sub):
    return importlib.import_module(PKG + "." + sub)


WORKLOADS = {
    # name: (batch per GPU, LR size, factor, description)
    "gan_x4": dict(batch=32, lr=128, factor=4, desc="full GAN step x4 (G + D + VGG19 perceptual), batch 32/GPU, 128x128->512x512"),
    "gen_l1_x4": dict(batch=16, lr=32, factor=4, desc="generator-only L1 step x4, batch 16/GPU, 32x32->128x128"),
    "infer_x8": dict(batch=1, lr=256, factor=8, desc="generator x8 eval forward, fp16, 256x256->2048x2048 (config 5)"),
    "dip_x2": dict(batch=1, lr=64, factor=2, desc="Deep-Image-Prior iteration x2, HR 128x128, fp16 storage (config 1 on the GPU)"),
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


def build_step(workload, dev, world):
    cfg = WORKLOADS[workload]
    Gm, optim, steps = P("models.GAN.generator"), P("optim"), P("steps")
    torch.manual_seed(0)                      # module-default init under seed 0
    if workload == "infer_x8":
        infer = P("infer")
        gen = Gm.Generator(8, 16).to(dev)
        img = torch.rand(1, 3, cfg["lr"], cfg["lr"], generator=torch.Generator().manual_seed(1)).to(dev)
        if os.environ.get("DSR_HIP_GRAPH", "1") != "0":      # ~150 small launches per image: replay them from a HIP graph
            return steps.GraphedStep(lambda: infer.super_resolve(gen, img)), (cfg["lr"] * 8) ** 2
        return (lambda: infer.super_resolve(gen, img)), (cfg["lr"] * 8) ** 2
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
        if os.environ.get("DSR_HIP_GRAPH", "1") != "0":
            return steps.GraphedStep(lambda: run.step()[0]), hr_sz * hr_sz
        return (lambda: run.step()[0]), hr_sz * hr_sz
    gen = Gm.Generator(cfg["factor"], 16).to(dev).train()
    n, s, f = cfg["batch"], cfg["lr"], cfg["factor"]
    g = torch.Generator(device="cpu").manual_seed(1 + (dist.get_rank() if world > 1 else 0))
    lr = torch.rand(n, 3, s, s, generator=g).to(dev)                     # LR ~ U(0,1)
    hr = (torch.rand(n, 3, s * f, s * f, generator=g) * 2 - 1).to(dev)   # HR ~ U(-1,1)
    D = P("dist")
    opt_g = optim.FusedAdam(gen.parameters(), lr=1e-4)
    sync_g = D.GradSync(gen.parameters()).attach()
    if workload == "gen_l1_x4":
        D.broadcast_module(gen)
        F = P("functional")

        def step():
            fake = gen(lr)
            loss = F.l1_loss(fake, hr)
            opt_g.zero_grad()
            loss.backward()
            sync_g()
            opt_g.step()
            return loss
        if world == 1 and os.environ.get("DSR_HIP_GRAPH", "1") != "0":
            # launch-bound workload (~500 launches per step): capture the whole step once, replay it per step
            return steps.GraphedStep(step), n * (s * f) ** 2
        return step, n * (s * f) ** 2
    Dm, GANu = P("models.GAN.discriminator"), P("utils.GAN")
    F = P("functional")
    disc = Dm.Discriminator((s * f, s * f)).to(dev).train()
    perc = GANu.PerceptualLoss().to(dev)
    D.broadcast_module(gen)
    D.broadcast_module(disc)
    opt_d = optim.FusedAdam(disc.parameters(), lr=1e-4)
    sync_d = D.GradSync(disc.parameters()).attach()

    def step():
        return steps.gan_step(gen, disc, perc, opt_g, opt_d, lr, hr, sync_g, sync_d,
                              overlap=os.environ.get("DSR_GAN_OVERLAP", "1") != "0")[1]
    step.modules = [gen, disc]
    # the roofline leg times kernels one at a time: on the single-stream form of the same step (identical launches,
    # identical arithmetic) a launch's HIP-event bracket is not stretched by kernels of the other stream
    step.serial = lambda: steps.gan_step(gen, disc, perc, opt_g, opt_d, lr, hr, sync_g, sync_d, overlap=False)[1]
    return step, n * (s * f) ** 2


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


def cpu_baseline(workload):
    """The oracle (CPU fp32 restatement of the reference) on a bounded sample of the same workload."""
    from oracle import dip, downsampler, filler, gan, recipes, vgg
    cores = host_cores()
    torch.set_num_threads(cores)
    cfg = WORKLOADS[workload]
    f = cfg["factor"]
    torch.manual_seed(0)
    if workload == "infer_x8":
        s = 128                                  # quarter-size image: the eval forward is linear in pixels
        gsd = filler.fill_state_dict(gan.template(gan.generator_shapes(8, 16)))
        x = torch.rand(1, 3, s, s)
        t0 = time.perf_counter()
        with torch.no_grad():
            gan.generator_forward(gsd, x, False)
        dt = time.perf_counter() - t0
        return {"value": (s * 8) ** 2 / dt / 1e6, "unit": "HR Mpixels/s", "cores": cores, "kind": "port",
                "sample": f"1 eval forward of a {s}x{s} LR image (workload: 256x256), no warm-up", "seconds_per_step": dt}
    if workload == "dip_x2":
        hr_sz = cfg["lr"] * 2
        dcfg = dip.SkipConfig(input_depth=32)
        st = recipes.DipState(filler.fill_state_dict(gan.template(dip.skip_shapes(dcfg))), dcfg,
                              torch.rand(1, 32, hr_sz, hr_sz) * 0.1, factor=2, lr=0.01, reg_noise_std=0.05)
        lr_img = downsampler.downsampler_forward(torch.rand(1, 3, hr_sz, hr_sz), 2, "lanczos2", phase=0.5, preserve_size=True)
        recipes.dip_step(st, lr_img)
        t0 = time.perf_counter()
        for _ in range(20):
            recipes.dip_step(st, lr_img)
        dt = (time.perf_counter() - t0) / 20
        return {"value": hr_sz * hr_sz / dt / 1e6, "unit": "HR Mpixels/s", "cores": cores, "kind": "port",
                "sample": "20 iterations after 1 warm-up", "seconds_per_step": dt}
    gsd = filler.fill_state_dict(gan.template(gan.generator_shapes(f, 16)))
    if workload == "gen_l1_x4":
        n, s, reps = cfg["batch"], cfg["lr"], 3
        st = recipes.GenOnlyState(gsd, lr=1e-4)
        lr, hr = torch.rand(n, 3, s, s), torch.rand(n, 3, s * f, s * f) * 2 - 1
        recipes.gen_l1_step(st, lr, hr)
        t0 = time.perf_counter()
        for _ in range(reps):
            recipes.gen_l1_step(st, lr, hr)
        dt = (time.perf_counter() - t0) / reps
        sample = f"{reps} full steps (batch {n}, {s}x{s}->{s*f}x{s*f}) after 1 warm-up"
    else:
        n, s, reps = 1, cfg["lr"], 1      # one sample of the batch-32 step: per-sample cost scales linearly in batch
        dsd = filler.fill_state_dict(gan.template(gan.discriminator_shapes((s * f, s * f))))
        vsd = filler.fill_state_dict({k: torch.zeros(v) for k, v in vgg.vgg_shapes().items()}, salt=3)
        st = recipes.GanState(gsd, dsd, vsd, lr=1e-4)
        lr, hr = torch.rand(n, 3, s, s), torch.rand(n, 3, s * f, s * f) * 2 - 1
        t0 = time.perf_counter()
        for _ in range(reps):
            recipes.gan_step(st, lr, hr)
        dt = (time.perf_counter() - t0) / reps
        sample = f"{reps} step at batch {n} of the batch-{cfg['batch']} workload ({s}x{s}->{s*f}x{s*f}), no warm-up"
    px = n * (s * f) ** 2
    return {"value": px / dt / 1e6, "unit": "HR Mpixels/s", "cores": cores, "kind": "port", "sample": sample,
            "seconds_per_step": dt}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="gan_x4", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
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
    note("timed region")
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    t_issue = time.perf_counter() - t0          # host time to ISSUE the steps (GPU still running): launch-bound if ~ dt
    fence()
    dt = time.perf_counter() - t0
    if world > 1 or force:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
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
        a.workload, "HR Mpixels/sec x4 GAN train step")
    out = {"metric": metric, "value": value, "unit": "HR Mpixels/s", "n_gpus": world,
           "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f16" if a.workload in ("infer_x8", "dip_x2") else "bf16", "data": "synthetic",
           "config": {"workload": a.workload + ": " + WORKLOADS[a.workload]["desc"],
                      "global_batch": WORKLOADS[a.workload]["batch"] * world, "parallelism": f"dp{world}"}}

    note(f"{ms:.2f} ms/step (host issue {t_issue / a.steps * 1e3:.2f} ms/step)")
    if rank == 0 and world == 1 and not a.no_roofline:
        note("roofline leg")
        F = P("functional")
        F.KERNEL_LOG = []
        getattr(step, "serial", step)()
        torch.cuda.synchronize()
        fam = {}
        for kind, d, e0, e1, k in F.KERNEL_LOG:
            t, fl, cnt = fam.get(k, (0.0, 0.0, 0))
            # one C-ABI call = one kernel launch, except a strided dgrad on the gather kernel (stride^2 parity classes)
            nl = d[7] * d[7] if (kind == "dgrad" and k.startswith("conv_gemm")) else 1
            fam[k] = (t + e0.elapsed_time(e1) * 1e-3, fl + conv_flops(d), cnt + nl)
        F.KERNEL_LOG = None
        if fam:
            top = max(fam, key=lambda k: fam[k][0])
            t, fl, cnt = fam[top]
            ach = fl / t / 1e12
            out["roofline"] = {"bound": "mfma", "kernel": top, "achieved": ach, "peak": MFMA_BF16_PEAK_TFLOPS,
                               "unit": "TFLOP/s", "frac": ach / MFMA_BF16_PEAK_TFLOPS,
                               "traffic": pmc_traffic(a.workload, top),
                               "launches": cnt, "avg_launch_ms": t / cnt * 1e3,
                               "measured_on": "single-stream form of the step (DSR_GAN_OVERLAP=0): per-kernel durations "
                                              "are not stretched by the concurrent D/G halves of the timed step",
                               "families": {k: {"seconds": v[0], "tflops": v[1] / v[0] / 1e12, "launches": v[2]}
                                            for k, v in fam.items()}}
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        note("cpu baseline (oracle on host cores)")
        out["cpu_baseline"] = cpu_baseline(a.workload)
    if rank == 0:
        print(json.dumps(out))
    if world > 1 or force:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
