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

from oracle import dip, downsampler, filler, gan, losses, lowp, recipes, vgg

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


from parity_util import compare_grads, cos, pre_bn_bias, prelu_ok  # noqa: E402  (tests/parity_util.py)


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
        # skipped: pre-BN conv biases (noise in the oracle) and the one-element PReLU slopes, whose gradient is a sum of
        # +/- terms over a whole activation map -- Adam turns a sign change of that near-cancelling sum into a full
        # +/- lr step, so their 10-step displacement is not a stable quantity under ANY storage rounding
        if float(d_ref.abs().max()) == 0.0 or pre_bn_bias(k) or p.numel() == 1:
            continue
        c = cos(d_hip, d_ref)
        if c < 0.90:
            bad.append((k, round(c, 4)))
    assert not bad, bad


# ----------------------------------------------------------------------------- config 1
def _dip_run(dev, lr_rate, iters, sim16=True):
    """DIP.py:47-99 at LR 64x64 -> HR 128x128, get_net(32,'skip','reflection',128,128,4,5,'bilinear'), reg_noise_std 0.05,
    with the per-iteration N(0,1) draw injected (same numbers on all sides).  Returns per-iteration
    (loss_hip, loss_ref, loss_sim, psnr_hip, psnr_ref, psnr_sim, psnr(hip, ref)) where `sim` is the fp32 oracle with
    fp16 conv storage (oracle/lowp.py): the product's storage policy restated with the oracle's arithmetic."""
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
    run = steps.DipRunner(net, down, zin.to(dev), lr_img.to(dev), lr_rate, 0.05)

    def state():
        return recipes.DipState({k: v.clone() for k, v in sd.items()}, cfg, zin.clone(), factor=2, lr=lr_rate,
                                reg_noise_std=0.05)
    st, sim = state(), state()
    trace = []
    for it in range(iters):
        noise = filler.tensor(f"in:c1_noise{it}", (1, 32, 128, 128), 1.7)   # variance ~ 1
        loss, out = run.step(noise.to(dev))
        rloss, rout = recipes.dip_step(st, lr_img, noise)
        if sim16:
            with lowp.storage(torch.float16):
                sloss, sout = recipes.dip_step(sim, lr_img, noise)
        else:
            sloss, sout = rloss, rout
        trace.append((loss.item(), rloss, sloss, losses.psnr(out.cpu(), hr, 1.0), losses.psnr(rout, hr, 1.0),
                      losses.psnr(sout, hr, 1.0), losses.psnr(out.cpu(), rout, 1.0)))
    return trace


def test_config1_dip_exact_size(dev):
    """Config 1 exactly (Adam lr 0.01, DIP.py:318), 20 iterations.  At this learning rate the trajectory is chaotic in its
    first ~10 iterations under ANY change of rounding: Adam's first updates are +/- lr per weight by the SIGN of a
    gradient, and the hourglass normalises 4x4 ... 8x8 maps at batch 1, so the fp32 oracle with nothing but fp16 conv
    storage already departs from itself by 16 % in the loss and to 12 dB between outputs before both settle on the same
    fit (measured here, recorded in gpurun_out/parity_baseline.json).  The bar is therefore stated against that floor:
    over the whole run, and again over the settled tail (iterations 15-19), the HIP path may deviate from the fp32 oracle
    at most 1.5x as far as the fp16-storage oracle does (+1 % / +0.5 % of the loss, +0.01 / +0.005 dB)."""
    trace = _dip_run(dev, 0.01, 20)
    dl_hip = [abs(t[0] - t[1]) / t[1] for t in trace]
    dl_sim = [abs(t[2] - t[1]) / t[1] for t in trace]
    dp_hip = [abs(t[3] - t[4]) for t in trace]
    record("config1", loss_rel_hip=dl_hip, loss_rel_fp16_storage_oracle=dl_sim, dpsnr_hip=dp_hip,
           dpsnr_fp16_storage_oracle=[abs(t[5] - t[4]) for t in trace], psnr_hip_vs_oracle_db=[t[6] for t in trace],
           loss_ref=[t[1] for t in trace])
    assert trace[-1][1] < 0.7 * trace[0][1] and trace[-1][0] < 0.7 * trace[0][0], trace    # the fit progresses on both sides
    assert dl_hip[0] <= 2e-3, trace[0]                             # first forward: identical weights and input
    dp_sim = [abs(t[5] - t[4]) for t in trace]
    # the yardstick itself is pinned: the fp16-storage floor of this trajectory (CPU only, deterministic) was measured at
    # 22.3 % (loss) -- a change of oracle/lowp.py that moves it out of the band fails here instead of moving the bars below
    assert 0.15 <= max(dl_sim) <= 0.30, max(dl_sim)
    assert max(dl_hip) <= 1.5 * max(dl_sim) + 0.01, (max(dl_hip), max(dl_sim))
    assert max(dp_hip) <= 1.5 * max(dp_sim) + 0.01, (max(dp_hip), max(dp_sim))
    # once the fit has settled (iterations 15-19); measured: HIP 3.9 % / 0.028 dB, fp16-storage oracle 5.5 % / 0.041 dB
    assert max(dl_hip[15:]) <= 1.5 * max(dl_sim[15:]) + 0.005, (dl_hip[15:], dl_sim[15:])
    assert max(dp_hip[15:]) <= 1.5 * max(dp_sim[15:]) + 0.005, (dp_hip[15:], dp_sim[15:])


def test_config1_dip_small_learning_rate(dev):
    """The same configuration at Adam lr 1e-4 (every other setting of config 1 unchanged), where the optimiser amplifies
    rounding far less: 20 iterations, |dPSNR| <= 0.02 dB and loss within 2 % at EVERY iteration (measured: 0.0145 dB,
    1.3 %; Adam still moves every weight by +/- lr per step on the sign of its gradient)."""
    trace = _dip_run(dev, 1e-4, 20, sim16=False)
    dl = [abs(t[0] - t[1]) / t[1] for t in trace]
    dp = [abs(t[3] - t[4]) for t in trace]
    record("config1_lr1e-4", loss_rel_hip=dl, dpsnr_hip=dp, psnr_hip_vs_oracle_db=[t[6] for t in trace])
    assert max(dl) <= 0.02, dl
    assert max(dp) <= 0.02, dp


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
    # tanh output in (-1,1) after 38 fp16-stored layers (measured: 0.039 max abs, 57.6 dB)
    assert e_whole <= 0.05 and e_tiled <= 0.05, (e_whole, e_tiled)
    assert p_whole >= 55.0 and p_tiled >= 55.0, (p_whole, p_tiled)
    assert e_wt <= 4e-3, e_wt          # same arithmetic; a tile's fp16 stores differ only where tile-local sums round differently


# ----------------------------------------------------------------------------- config 3 (full spatial size, batch 2)
@pytest.fixture(scope="module")
def c3_states():
    """Closed-form parameters of Generator(4,16) and Discriminator((512,512)) (541.6 M values), filled once."""
    gsd = filler.fill_state_dict(gan.template(gan.generator_shapes(4, 16)))
    dsd = filler.fill_state_dict(gan.template(gan.discriminator_shapes((512, 512))))
    return gsd, dsd


@pytest.fixture(scope="module")
def c3_oracle(c3_states):
    """Step 1 of the config-3 recipe on the CPU, twice: the fp32 oracle, and the same oracle with bf16 conv storage
    (oracle/lowp.py) -- the FLOOR any bf16-storage implementation sits on.  Measured (this file's record()): through 33
    train-mode BatchNorms at batch 2 the idealised bf16 model already moves early-layer gradient tensors to cosine
    0.946-0.97 against fp32 and flips the sign of near-cancelling PReLU-slope sums."""
    GANu = P("utils.GAN")
    gsd, dsd = c3_states
    vsd = GANu._standin_vgg_state()
    lr = filler.tensor("in:c3_lr", (2, 3, 128, 128), 0.5, 0.5)
    hr = filler.tensor("in:c3_hr", (2, 3, 512, 512))
    sim_cap = {}
    st = recipes.GanState({k: v.clone() for k, v in gsd.items()}, {k: v.clone() for k, v in dsd.items()}, vsd, lr=1e-4)
    with lowp.storage(torch.bfloat16):
        recipes.gan_step(st, lr, hr, capture=sim_cap)
    del st
    # ... and the fp32 oracle's two steps themselves (deterministic; both stream forms of the HIP step are compared with them)
    st = recipes.GanState({k: v.clone() for k, v in gsd.items()}, {k: v.clone() for k, v in dsd.items()}, vsd, lr=1e-4)
    ref_steps = []
    for _ in range(2):
        cap = {}
        rld, rlg, rfake = recipes.gan_step(st, lr, hr, capture=cap)
        ref_steps.append((rld, rlg, rfake, {k: cap[k] for k in ("g_grads", "d_grads")} if not ref_steps else None))
    return vsd, lr, hr, sim_cap, ref_steps, st


@pytest.mark.parametrize("overlap,big_tile", [(True, False), (False, False), (False, True)],
                         ids=["two_stream", "single_stream", "single_stream_256x256_tiles"])
def test_config3_gan_step_full_spatial_size(dev, c3_states, c3_oracle, overlap, big_tile, monkeypatch):
    """train_GAN.py:38-71 at LR 128x128 -> HR 512x512 with Discriminator((512,512)) (537 M-weight dense1) and the
    256/224 VGG preprocessing, batch 2, two steps, both the two-stream and the single-stream form of the step.
    Per step: loss_D and loss_G within 2 %, |dPSNR| <= 0.02 dB.  After step 1 every gradient tensor the two Adam steps
    consumed (G: content loss, D: loss_D) is compared with the fp32 oracle by cosine and norm ratio, against the
    bf16-storage floor of the SAME computation (c3_oracle) by the rule of tests/parity_util.py.  After step 2: BatchNorm
    running statistics (1 %) and counters.
    Third form: at batch 2 no layer has the >= 150 tiles at which the dispatcher takes the 256x256 tile of the gather kernel
    (the dominant kernel of the batch-32 step), so DSR_CONV_BIG_TILES=1 sends EVERY layer with N % 256 == 0 (D.b3 ... b6
    forward, D.b5 input gradient, VGG conv3_1 ... conv5_4 forward and input gradients) through it -- including its stream-K
    form for launches that do not fill whole rounds of the chip -- against the same oracle numbers."""
    if big_tile:
        monkeypatch.setenv("DSR_CONV_BIG_TILES", "1")
        import ctypes as C
        L = P("_lib")
        dsc = L.ConvDesc(L.BF16, 2, 128, 128, 128, 256, 3, 3, 1, 1, 0)           # D.b3 at this test's batch
        # (with its BatchNorm statistics epilogue the layer takes the 256x256 tile; a launch without one may take the 224x256 form)
        assert "x256>" in L.lib().dsr_conv_kernel_name(C.byref(dsc), 0, None).decode()
    Gm, Dm, GANu, optim, steps = (P("models.GAN.generator"), P("models.GAN.discriminator"), P("utils.GAN"), P("optim"),
                                  P("steps"))
    n, s, f = 2, 128, 4
    gsd, dsd = c3_states
    vsd, lr, hr, sim_cap, ref_steps, st = c3_oracle
    g, d = Gm.Generator(f, 16), Dm.Discriminator((s * f, s * f))
    g.load_state_dict(gsd), d.load_state_dict(dsd)
    g.to(dev).train(), d.to(dev).train()
    perc = GANu.PerceptualLoss().to(dev)
    og, od = optim.FusedAdam(g.parameters(), lr=1e-4), optim.FusedAdam(d.parameters(), lr=1e-4)
    lrd, hrd = lr.to(dev), hr.to(dev)
    rec = {}
    for it in range(2):
        ld, lg, fake = steps.gan_step(g, d, perc, og, od, lrd, hrd, overlap=overlap)
        torch.cuda.synchronize()
        rld, rlg, rfake, cap = ref_steps[it]
        dp = abs(losses.psnr(fake.cpu(), hr) - losses.psnr(rfake, hr))
        rec[f"step{it}"] = dict(loss_d=(ld.item(), rld), loss_g=(lg.item(), rlg), dpsnr_db=dp,
                                psnr_hip_vs_oracle_db=losses.psnr(fake.cpu(), rfake))
        assert dp <= 0.02, rec
        assert abs(ld.item() - rld) <= 0.02 * max(abs(rld), 0.1), rec
        assert abs(lg.item() - rlg) <= 0.02 * max(abs(rlg), 0.1), rec
        if it == 0:
            hip_g = dict((k, p.grad) for k, p in g.named_parameters())
            hip_d = dict((k, p.grad) for k, p in d.named_parameters())
            bad_g, tab_g = compare_grads(hip_g, cap["g_grads"], sim_cap["g_grads"], "G:")
            bad_d, tab_d = compare_grads(hip_d, cap["d_grads"], sim_cap["d_grads"], "D:")
            table = {**tab_g, **tab_d}
            worst = sorted(table.items(), key=lambda kv: kv[1][0])[:8]
            rec["grad_worst (cos_hip, cos_bf16_floor, ratio_hip, ratio_floor, numel)"] = worst
            rec["grad_min_cos_hip"] = worst[0][1][0]
            rec["grad_min_cos_floor"] = min(v[1] for v in table.values())
            rec["grad_tensors_compared"] = len(table)
            record(f"config3_overlap{int(overlap)}_big{int(big_tile)}", **rec)
            # the yardstick itself is pinned: the bf16-storage floor of this computation (CPU only, deterministic) must sit in
            # the band it was measured in, so that a change of oracle/lowp.py fails HERE instead of silently moving the bars
            assert 0.92 <= rec["grad_min_cos_floor"] <= 0.97, rec["grad_min_cos_floor"]
            assert not (bad_g + bad_d), bad_g + bad_d
            assert not prelu_ok(hip_g, cap["g_grads"], sim_cap["g_grads"])
    record(f"config3_overlap{int(overlap)}_big{int(big_tile)}", **rec)
    for mod, osd in ((g, st.g), (d, st.d)):
        for k, v in mod.state_dict().items():
            if "running_" in k:
                r = osd[k]
                assert float((v.cpu() - r).abs().max() / r.abs().max()) < 1e-2, k
            if "num_batches" in k:
                assert int(v) == int(osd[k]), k
    assert all(int(v) == 4 for k, v in g.state_dict().items() if k.endswith("num_batches_tracked"))
    assert all(int(v) == 6 for k, v in d.state_dict().items() if k.endswith("num_batches_tracked"))
