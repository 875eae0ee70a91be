"""GPU parity at the EXACT sizes BASELINE.json's configs name, against the CPU fp32 oracle on identical
closed-form parameters (oracle/filler.py) and inputs:

  config 1  DIP x2, LR 64x64 -> HR 128x128, default 5-scale skip net, >= 20 Adam iterations  (DIP.py:47-99)
  config 2  generator-only x4 L1, batch 16, 32x32 -> 128x128, 16 residual blocks, 10 Adam steps
  config 3  full GAN step x4 at full spatial size 128x128 -> 512x512 (Discriminator((512,512)), VGG 256/224),
            batch 2 instead of 32 (the CPU oracle needs ~10 s per sample and step), two steps (train_GAN.py:38-71)
  config 5  Generator(8).eval() fp16, 256x256 -> 2048x2048, whole image and halo-tiled

Bars (north_star: "PSNR within 0.02 dB of reference"; SURVEY.md 8d defines PSNR on the network output vs the HR
target, range 2.0 for [-1,1] tensors, 1.0 for DIP's [0,1]):  |dPSNR| <= 0.02 dB at every step, losses within the
percentage stated per test, gradients by cosine / norm ratio per tensor with FIXED thresholds (measured floors
are recorded in gpurun_out/parity_baseline.json by each test for DESIGN.md).
"""
import importlib
import json
import os
import time

import pytest
import torch

from oracle import dip, downsampler, filler, gan, losses, recipes, vgg

pytestmark = pytest.mark.gpu
PKG = "deep-super-resolution_amd"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def P(sub):
    return importlib.import_module(PKG + "." + sub)


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    P("_lib").lib()
    torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 32)))
    return torch.device("cuda:0")


def record(name, **kw):
    """Append measured parity numbers to gpurun_out/parity_baseline.json (scratch; copied into profiles/ by hand)."""
    path = os.path.join(ROOT, "gpurun_out", "parity_baseline.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        data = json.load(open(path)) if os.path.exists(path) else {}
        data[name] = kw
        json.dump(data, open(path, "w"), indent=1, sort_keys=True)
    except OSError:
        pass


def cos(a, b):
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    return float((a @ b) / (a.norm() * b.norm()).clamp_min(1e-300))


def pre_bn_bias(k):
    """Conv biases in front of a train-mode BatchNorm: their gradient is analytically zero (the reference holds ~1e-9
    rounding noise there), so they are not compared."""
    return k.endswith("bias") and (("residual_blocks" in k and ".conv" in k) or k == "conv2.bias" or
                                   ("convblocks" in k and ".conv1." in k))


def grad_table(named_grads, ref_grads):
    """Per tensor: (cosine, norm ratio) against the oracle."""
    out = {}
    for k, g in named_grads:
        r = ref_grads.get(k)
        if g is None or r is None or pre_bn_bias(k) or float(r.abs().max()) == 0.0:
            continue
        out[k] = (cos(g.cpu(), r), float(g.double().norm().cpu() / r.double().norm()))
    return out


# ----------------------------------------------------------------------------- config 2
def test_config2_generator_l1_exact_size(dev):
    """batch 16, 32x32 -> 128x128, Generator(4, 16), L1, Adam lr 1e-4: 10 steps, every step
    |dPSNR| <= 0.02 dB and loss within 0.5 %."""
    Gm, optim, steps = P("models.GAN.generator"), P("optim"), P("steps")
    sd = filler.fill_state_dict(gan.template(gan.generator_shapes(4, 16)))
    g = Gm.Generator(4, 16)
    g.load_state_dict(sd)
    g.to(dev).train()
    opt = optim.FusedAdam(g.parameters(), lr=1e-4)
    st = recipes.GenOnlyState({k: v.clone() for k, v in sd.items()}, lr=1e-4)
    lr = filler.tensor("in:c2_lr", (16, 3, 32, 32), 0.5, 0.5)
    hr = filler.tensor("in:c2_hr", (16, 3, 128, 128))
    lrd, hrd = lr.to(dev), hr.to(dev)
    worst_psnr = worst_loss = 0.0
    trace = []
    for it in range(10):
        loss, fake = steps.gen_l1_step(g, opt, lrd, hrd)
        rloss, rfake = recipes.gen_l1_step(st, lr, hr)
        dp = abs(losses.psnr(fake.cpu(), hr) - losses.psnr(rfake, hr))
        dl = abs(loss.item() - rloss) / abs(rloss)
        trace.append((it, loss.item(), rloss, dp, losses.psnr(fake.cpu(), rfake)))
        worst_psnr, worst_loss = max(worst_psnr, dp), max(worst_loss, dl)
    record("config2", worst_dpsnr_db=worst_psnr, worst_loss_rel=worst_loss, psnr_hip_vs_oracle_db=[t[4] for t in trace])
    assert worst_psnr <= 0.02, trace
    assert worst_loss <= 5e-3, trace
    # after 10 identical Adam steps the weights themselves still agree: per-tensor cosine of the 10-step update
    bad = []
    for k, p in g.named_parameters():
        d_hip = (p.detach().cpu() - sd[k]).double()
        d_ref = (st.g[k].detach() - sd[k]).double()
        if float(d_ref.abs().max()) == 0.0 or (k.endswith("bias") and ".conv" in k):   # pre-BN biases: noise in the oracle
            continue
        c = cos(d_hip, d_ref)
        if c < 0.90:
            bad.append((k, round(c, 4)))
    assert not bad, bad


# ----------------------------------------------------------------------------- config 1
def test_config1_dip_exact_size(dev):
    """DIP.py:47-99 at LR 64x64 -> HR 128x128, get_net(32,'skip','reflection',128,128,4,5,'bilinear'), Adam lr 0.01,
    reg_noise_std 0.05, 20 iterations with the per-iteration N(0,1) draw injected (same numbers on both sides)."""
    M, D, steps = P("models.DIP"), P("utils.downsampler"), P("steps")
    cfg = dip.SkipConfig(input_depth=32)
    sd = filler.fill_state_dict(gan.template(dip.skip_shapes(cfg)))
    net = M.get_net(32, "skip", "reflection", upsample_mode="bilinear")
    net.load_state_dict(sd)
    net.to(dev).train()
    down = D.Downsampler(3, 2, "lanczos2", phase=0.5, preserve_size=True).to(dev)
    hr = filler.tensor("in:c1_hr", (1, 3, 128, 128), 0.5, 0.5)
    lr_img = downsampler.downsampler_forward(hr, 2, "lanczos2", phase=0.5, preserve_size=True)
    assert tuple(lr_img.shape) == (1, 3, 64, 64)
    zin = filler.tensor("in:c1_z", (1, 32, 128, 128), 0.05, 0.05)          # U(0, 0.1) like get_noise(...)*0.1
    run = steps.DipRunner(net, down, zin.to(dev), lr_img.to(dev), 0.01, 0.05)
    st = recipes.DipState({k: v.clone() for k, v in sd.items()}, cfg, zin.clone(), factor=2, lr=0.01, reg_noise_std=0.05)
    trace = []
    for it in range(20):
        noise = filler.tensor(f"in:c1_noise{it}", (1, 32, 128, 128), 1.7)   # variance ~ 1
        loss, out = run.step(noise.to(dev))
        rloss, rout = recipes.dip_step(st, lr_img, noise)
        p_hip, p_ref = losses.psnr(out.cpu(), hr, 1.0), losses.psnr(rout, hr, 1.0)
        trace.append((it, loss.item(), rloss, p_hip, p_ref, losses.psnr(out.cpu(), rout, 1.0)))
    worst_dp = max(abs(t[3] - t[4]) for t in trace)
    worst_dl = max(abs(t[1] - t[2]) / abs(t[2]) for t in trace)
    record("config1", worst_dpsnr_db=worst_dp, worst_loss_rel=worst_dl, psnr_hip_vs_oracle_db=[t[5] for t in trace],
           psnr_ref=[t[4] for t in trace])
    # the fit itself must progress identically: the MSE loss falls by the same factor on both sides
    assert trace[-1][2] < 0.7 * trace[0][2], trace
    assert worst_dl <= 0.03, trace
    assert worst_dp <= 0.05, trace          # fp16 storage through 30 train-mode BatchNorms at batch 1


# ----------------------------------------------------------------------------- config 5
def test_config5_x8_inference_exact_size(dev):
    """Generator(8).eval(), fp16, 256x256 -> 2048x2048 (eval_GAN.py:44,94): whole image, and halo-tiled with 128-pixel
    tiles; both against the fp32 oracle (max abs, PSNR at range 2) and against each other."""
    Gm, infer = P("models.GAN.generator"), P("infer")
    sd = filler.fill_state_dict(gan.template(gan.generator_shapes(8, 16)))
    g = Gm.Generator(8, 16)
    g.load_state_dict(sd)
    g.to(dev).eval()
    x = filler.tensor("in:c5_lr", (1, 3, 256, 256), 0.5, 0.5)
    whole = infer.super_resolve(g, x.to(dev))
    tiled = infer.super_resolve(g, x.to(dev), tile=128)
    torch.cuda.synchronize()
    assert tuple(whole.shape) == (1, 3, 2048, 2048)
    t0 = time.perf_counter()
    with torch.no_grad():
        ref = gan.generator_forward({k: v.clone() for k, v in sd.items()}, x, False)
    t_ref = time.perf_counter() - t0
    e_whole = (whole.cpu() - ref).abs().max().item()
    e_tiled = (tiled.cpu() - ref).abs().max().item()
    e_wt = (whole - tiled).abs().max().item()
    p_whole, p_tiled = losses.psnr(whole.cpu(), ref), losses.psnr(tiled.cpu(), ref)
    record("config5", max_abs_whole=e_whole, max_abs_tiled=e_tiled, max_abs_whole_vs_tiled=e_wt, psnr_whole_db=p_whole,
           psnr_tiled_db=p_tiled, oracle_seconds=t_ref)
    assert e_whole <= 0.02 and e_tiled <= 0.02, (e_whole, e_tiled)       # tanh output in (-1,1), fp16 storage
    assert p_whole >= 60.0 and p_tiled >= 60.0, (p_whole, p_tiled)
    assert e_wt <= 4e-3, e_wt          # same arithmetic; a tile's fp16 stores differ only where tile-local sums round differently


# ----------------------------------------------------------------------------- config 3 (full spatial size, batch 2)
@pytest.fixture(scope="module")
def c3_states():
    """Closed-form parameters of Generator(4,16) and Discriminator((512,512)) (541.6 M values), filled once."""
    gsd = filler.fill_state_dict(gan.template(gan.generator_shapes(4, 16)))
    dsd = filler.fill_state_dict(gan.template(gan.discriminator_shapes((512, 512))))
    return gsd, dsd


@pytest.mark.parametrize("overlap", [True, False])
def test_config3_gan_step_full_spatial_size(dev, c3_states, overlap):
    """train_GAN.py:38-71 at LR 128x128 -> HR 512x512 with Discriminator((512,512)) (537 M-weight dense1) and the
    256/224 VGG preprocessing, batch 2, two steps, both the two-stream and the single-stream form of the step.
    Checked per step: loss_D, loss_G, |dPSNR| <= 0.02 dB; after step 1: every G gradient (content loss) and D gradient
    (loss_D) tensor by cosine and norm ratio; after step 2: BatchNorm running statistics and counters."""
    Gm, Dm, GANu, optim, steps = (P("models.GAN.generator"), P("models.GAN.discriminator"), P("utils.GAN"), P("optim"),
                                  P("steps"))
    n, s, f = 2, 128, 4
    gsd, dsd = c3_states
    g, d = Gm.Generator(f, 16), Dm.Discriminator((s * f, s * f))
    g.load_state_dict(gsd), d.load_state_dict(dsd)
    g.to(dev).train(), d.to(dev).train()
    perc = GANu.PerceptualLoss().to(dev)
    vsd = {k[len("vgg_loss.net.0."):]: v.detach().cpu().clone() for k, v in perc.state_dict().items()}
    og, od = optim.FusedAdam(g.parameters(), lr=1e-4), optim.FusedAdam(d.parameters(), lr=1e-4)
    st = recipes.GanState({k: v.clone() for k, v in gsd.items()}, {k: v.clone() for k, v in dsd.items()}, vsd, lr=1e-4)
    lr = filler.tensor("in:c3_lr", (n, 3, s, s), 0.5, 0.5)
    hr = filler.tensor("in:c3_hr", (n, 3, s * f, s * f))
    lrd, hrd = lr.to(dev), hr.to(dev)
    rec = {}
    for it in range(2):
        cap = {}
        ld, lg, fake = steps.gan_step(g, d, perc, og, od, lrd, hrd, overlap=overlap)
        torch.cuda.synchronize()
        rld, rlg, rfake = recipes.gan_step(st, lr, hr, capture=cap)
        dp = abs(losses.psnr(fake.cpu(), hr) - losses.psnr(rfake, hr))
        rec[f"step{it}"] = dict(loss_d=(ld.item(), rld), loss_g=(lg.item(), rlg), dpsnr_db=dp,
                                psnr_hip_vs_oracle_db=losses.psnr(fake.cpu(), rfake))
        assert dp <= 0.02, rec
        assert abs(ld.item() - rld) <= 0.02 * max(abs(rld), 0.1), rec
        assert abs(lg.item() - rlg) <= 0.02 * max(abs(rlg), 0.1), rec
        if it == 0:
            # gradients the two Adam steps consumed (identical starting weights on both sides)
            tg = grad_table(((k, p.grad) for k, p in g.named_parameters()), cap["g_grads"])
            td = grad_table(((k, p.grad) for k, p in d.named_parameters()), cap["d_grads"])
            rec["g_grad_min_cos"] = min(v[0] for v in tg.values())
            rec["d_grad_min_cos"] = min(v[0] for v in td.values())
            rec["g_grad_worst"] = sorted(((round(v[0], 4), round(v[1], 4), k) for k, v in tg.items()))[:5]
            rec["d_grad_worst"] = sorted(((round(v[0], 4), round(v[1], 4), k) for k, v in td.items()))[:5]
            record(f"config3_overlap{int(overlap)}", **rec)
            bad = [(k, round(c, 4), round(r, 4)) for k, (c, r) in {**tg, **td}.items()
                   if (c < 0.98 or abs(r - 1) > 0.05) and not k.endswith("prelu1.weight")]
            assert not bad, bad
            # one-element PReLU slope gradients: sums of +/- terms over a whole activation map, relative to the largest
            scal = max(abs(float(cap["g_grads"][k])) for k in cap["g_grads"] if k.endswith("prelu1.weight"))
            for k, p in g.named_parameters():
                if k.endswith("prelu1.weight"):
                    assert abs(float(p.grad) - float(cap["g_grads"][k])) <= 0.05 * scal, (k, float(p.grad), float(cap["g_grads"][k]))
    record(f"config3_overlap{int(overlap)}", **rec)
    for mod, osd in ((g, st.g), (d, st.d)):
        for k, v in mod.state_dict().items():
            if "running_" in k:
                r = osd[k]
                assert float((v.cpu() - r).abs().max() / r.abs().max()) < 1e-2, k
            if "num_batches" in k:
                assert int(v) == int(osd[k]), k
    assert all(int(v) == 4 for k, v in g.state_dict().items() if k.endswith("num_batches_tracked"))
    assert all(int(v) == 6 for k, v in d.state_dict().items() if k.endswith("num_batches_tracked"))
