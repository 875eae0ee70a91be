"""CPU: the oracle restatement vs golden vectors captured from the reference's own modules
(tests/golden/make_golden.py).  This is what pins the oracle (prompt section 3)."""
import numpy as np
import pytest
import torch

from oracle import dip, downsampler, filler, gan, losses, recipes, vgg

torch.backends.mkldnn.enabled = False
# Goldens were computed by the reference modules in float64 and stored rounded to float32, so the
# oracle is run in float64 here and must agree to float32 rounding: the pin is formula-exact.
DT = torch.float64
TOL = dict(rtol=2e-6, atol=1e-7)


def T(name, shape, scale=1.0, offset=0.0):
    return filler.tensor(name, shape, scale, offset, dtype=DT)


def filled(shapes, salt=0):
    t = {k: (v if k.endswith("num_batches_tracked") else v.to(DT)) for k, v in gan.template(shapes).items()}
    return filler.fill_state_dict(t, salt)


def close_rel(got, ref, rel):
    """max |got-ref| <= rel * max |ref|: the pin for gradients, whose fp32 noise through small-batch
    BatchNorm backward is proportional to the tensor's scale, not to each element."""
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    err = np.abs(got - ref).max()
    assert err <= rel * max(np.abs(ref).max(), 1e-12), (err, np.abs(ref).max())


def probe_backward(y, name):
    (y * T(name, tuple(y.shape))).sum().backward()


def gsum(t):
    g = t.grad.detach().double().reshape(-1)
    return np.array([g.sum().item(), g.abs().sum().item(), g[0].item(), g[-1].item()])


def pre_bn_bias(sd, k):
    """Conv biases followed by a train-mode BatchNorm: analytically zero gradient, the reference holds
    rounding noise there (hard parts) -- not a parity quantity."""
    import re
    if re.fullmatch(r"(1\.1\.7\.)*2\.bias", k):
        # DIP: BN(skip+k) shift feeds reflect-pad conv -> BN: a per-channel constant, removed again
        return True
    if not k.endswith(".bias") or sd[k[:-4] + "weight"].dim() != 4:
        return False
    if k in ("conv1.bias", "conv3.bias", "conv.bias") or k.startswith("pixel_shuffle_blocks") or k.startswith("9."):
        return False
    return True


def check_gsums(sd, z, scale_tol=1e-7):
    for k in gan.trainable(sd):
        if pre_bn_bias(sd, k):
            continue
        ref = z["gsum/" + k]
        got = gsum(sd[k])
        # abs-sum is the robust pin; sum/first/last can be ~0 (pre-BN biases: analytically zero grads)
        assert abs(got[1] - ref[1]) <= scale_tol * max(ref[1], 1e-4) + 1e-9, (k, got, ref)
        if ref[1] > 1e-3:
            assert abs(got[0] - ref[0]) <= scale_tol * ref[1] + 1e-9, (k, got, ref)
            assert abs(got[2] - ref[2]) <= 1e-6 * max(abs(ref[2]), ref[1] / sd[k].numel()) + 1e-9, (k, got, ref)


def test_resblock(golden):
    z = golden("resblock")
    shapes = {k[len("residual_blocks.0."):]: v for k, v in gan.generator_shapes(8, 1).items()
              if k.startswith("residual_blocks.0.")}
    sd = filled(shapes)
    sd = {"rb." + k: v for k, v in sd.items()}
    # filler keys must match the reference module's own key names (no prefix)
    sd = {"rb." + k: v for k, v in filled(shapes).items()}
    params = recipes.leaves(sd, [k for k in sd if k in gan.trainable(sd)])
    x = T("in:resblock", (2, 64, 8, 8)).requires_grad_(True)
    y = gan.residual_block(sd, "rb", x, True)
    probe_backward(y, "probe:resblock")
    np.testing.assert_allclose(y.detach().numpy(), z["y_train"], **TOL)
    np.testing.assert_allclose(x.grad.numpy(), z["gx_train"], rtol=2e-6, atol=1e-7)
    for k in gan.trainable(sd):
        np.testing.assert_allclose(sd[k].grad.numpy(), z["grad/" + k[3:]], rtol=2e-6, atol=1e-7)
    for k in sd:
        if "running_" in k or "num_batches" in k:
            np.testing.assert_allclose(sd[k].numpy(), z["buf/" + k[3:]], rtol=2e-6, atol=1e-7)
    ye = gan.residual_block(sd, "rb", x.detach(), False)
    np.testing.assert_allclose(ye.detach().numpy(), z["y_eval"], **TOL)
    del params


def test_psblock(golden):
    z = golden("psblock")
    sd = filled({"conv1.weight": (256, 64, 3, 3), "conv1.bias": (256,), "prelu1.weight": (1,)})
    recipes.leaves(sd)
    x = T("in:psblock", (2, 64, 6, 6)).requires_grad_(True)
    y = torch.nn.functional.conv2d(x, sd["conv1.weight"], sd["conv1.bias"], padding=1)
    y = gan.prelu(gan.pixel_shuffle2(y), sd["prelu1.weight"])
    probe_backward(y, "probe:psblock")
    np.testing.assert_allclose(y.detach().numpy(), z["y"], **TOL)
    np.testing.assert_allclose(x.grad.numpy(), z["gx"], rtol=2e-6, atol=1e-7)
    for k in sd:
        np.testing.assert_allclose(sd[k].grad.numpy(), z["grad/" + k], rtol=2e-6, atol=1e-7)


@pytest.mark.parametrize("tag,factor,nres", [("g8_r2", 8, 2), ("g8_r16", 8, 16), ("g4_r16", 4, 16),
                                             ("g4_r2", 4, 2), ("g2_r1", 2, 1)])
def test_generator(golden, tag, factor, nres):
    z = golden("generator_" + tag)
    sd = filled(gan.generator_shapes(factor, nres))
    recipes.leaves(sd)
    x = T("in:" + tag, (2, 3, 8, 8), 0.5, 0.5).requires_grad_(True)
    y = gan.generator_forward(sd, x, True)
    assert y.shape == (2, 3, 8 * factor, 8 * factor)
    probe_backward(y, "probe:" + tag)
    np.testing.assert_allclose(y.detach().numpy(), z["y_train"], rtol=2e-6, atol=1e-7)
    close_rel(x.grad.numpy(), z["gx_train"], 1e-6)
    check_gsums(sd, z)
    for k in sd:
        if "running_" in k or "num_batches" in k:
            np.testing.assert_allclose(sd[k].numpy(), z["buf1/" + k], rtol=2e-6, atol=1e-7)
    gan.generator_forward(sd, x.detach(), True)
    for k in sd:
        if "running_" in k or "num_batches" in k:
            np.testing.assert_allclose(sd[k].numpy(), z["buf2/" + k], rtol=2e-6, atol=1e-7)
    ye = gan.generator_forward(sd, x.detach(), False)
    np.testing.assert_allclose(ye.detach().numpy(), z["y_eval"], rtol=2e-6, atol=1e-7)


@pytest.mark.parametrize("hw", [(32, 32), (64, 64), (48, 32)])
def test_discriminator(golden, hw):
    tag = f"d_{hw[0]}x{hw[1]}"
    z = golden("discriminator_" + tag)
    sd = filled(gan.discriminator_shapes(hw))
    recipes.leaves(sd)
    x = T("in:" + tag, (3, 3, hw[0], hw[1])).requires_grad_(True)
    y = gan.discriminator_forward(sd, x, True)
    probe_backward(y, "probe:" + tag)
    np.testing.assert_allclose(y.detach().numpy(), z["y_train"], rtol=2e-6, atol=1e-7)
    close_rel(x.grad.numpy(), z["gx_train"], 1e-6)
    check_gsums(sd, z)
    for k in sd:
        if "running_" in k or "num_batches" in k:
            np.testing.assert_allclose(sd[k].numpy(), z["buf/" + k], rtol=2e-6, atol=1e-7)
    ye = gan.discriminator_forward(sd, x.detach(), False)
    np.testing.assert_allclose(ye.detach().numpy(), z["y_eval"], rtol=2e-6, atol=1e-7)


DIP_CASES = [("sq64", (1, 32, 64, 64), {}), ("r64x96", (1, 32, 64, 96), {}), ("crop72x104", (1, 32, 72, 104), {}),
             ("b2_64", (2, 32, 64, 64), {}),
             ("small", (1, 8, 32, 32), dict(skip_n33d=16, skip_n33u=16, skip_n11=4, num_scales=3))]
_S3 = dict(skip_n33d=16, skip_n33u=16, skip_n11=4, num_scales=3)
# get_net's optional arguments (act_fun / downsample_mode / upsample_mode), same cases as make_golden.DIP_OPTION_CASES
DIP_CASES += [
    ("opt_elu_avg_nearest", (1, 8, 32, 32), dict(_S3, act_fun="ELU", downsample_mode="avg", upsample_mode="nearest")),
    ("opt_none_max_bilinear", (1, 8, 32, 32), dict(_S3, act_fun="none", downsample_mode="max", upsample_mode="bilinear")),
    ("opt_leaky_avg_nearest_odd", (1, 8, 36, 44), dict(_S3, act_fun="LeakyReLU", downsample_mode="avg",
                                                       upsample_mode="nearest")),
    ("opt_elu_stride_nearest_b2", (2, 8, 32, 48), dict(_S3, act_fun="ELU", downsample_mode="stride",
                                                       upsample_mode="nearest")),
]


@pytest.mark.parametrize("tag,shape,kw", DIP_CASES)
def test_dip(golden, tag, shape, kw):
    z = golden("dip_" + tag)
    cfg = dip.SkipConfig(input_depth=shape[1], **kw)
    shapes = dip.skip_shapes(cfg)
    assert set(shapes) == set(str(k) for k in z["keys"])          # state_dict key names (1-indexed paths)
    sd = filled(shapes)
    recipes.leaves(sd)
    x = T("in:dip_" + tag, shape, 0.05, 0.05).requires_grad_(True)
    y = dip.skip_forward(sd, x, cfg, True)
    probe_backward(y, "probe:dip_" + tag)
    np.testing.assert_allclose(y.detach().numpy(), z["y"], rtol=2e-6, atol=1e-7)
    close_rel(x.grad.numpy()[:, :4], z["gx"], 1e-6)
    check_gsums(sd, z)
    for k in sd:
        if "running_" in k or "num_batches" in k:
            np.testing.assert_allclose(sd[k].numpy(), z["buf/" + k], rtol=2e-6, atol=1e-7)


def test_downsampler(golden):
    z = golden("downsampler")
    for f in (2, 4, 8):
        kt, kw, sup, sig = downsampler.resolve(f, "lanczos2")
        k = downsampler.get_kernel(f, kt, 0.5, kw, support=sup, sigma=sig)
        assert k.shape == (4 * f, 4 * f)
        np.testing.assert_allclose(k, z[f"kernel_f{f}"], rtol=1e-12, atol=1e-15)
        assert downsampler.padding_of(k.shape[0], f) == int(z[f"pad_f{f}"][0])
        x = T(f"in:down{f}", (1, 3, 32, 32), 0.5, 0.5).requires_grad_(True)
        y = downsampler.downsampler_forward(x, f, "lanczos2", phase=0.5, preserve_size=True)
        assert y.shape == (1, 3, 32 // f, 32 // f)
        probe_backward(y, f"probe:down{f}")
        np.testing.assert_allclose(y.detach().numpy(), z[f"y_f{f}"], rtol=2e-6, atol=1e-7)
        np.testing.assert_allclose(x.grad.numpy(), z[f"gx_f{f}"], rtol=2e-6, atol=1e-7)
    x = T("in:downl3", (2, 3, 16, 20), 0.5, 0.5)
    for tag, args in (("l3", dict(factor=2, kernel_type="lanczos3", phase=0, preserve_size=True)),
                      ("g12", dict(factor=2, kernel_type="gauss12", phase=0, preserve_size=True)),
                      ("box", dict(factor=4, kernel_type="box", phase=0.5, kernel_width=4, preserve_size=False))):
        kt, kw, sup, sig = downsampler.resolve(args["factor"], args["kernel_type"], args.get("kernel_width"))
        k = downsampler.get_kernel(args["factor"], kt, args["phase"], kw, support=sup, sigma=sig)
        np.testing.assert_allclose(k, z["kernel_" + tag], rtol=1e-12, atol=1e-15)
        y = downsampler.downsampler_forward(x, **args)
        np.testing.assert_allclose(y.numpy(), z["y_" + tag], rtol=2e-6, atol=1e-7)


def test_get_noise(golden):
    torch.manual_seed(0)
    t = dip.get_noise(32, (8, 12))          # float32, torch CPU generator
    np.testing.assert_array_equal(t.numpy(), golden("get_noise_seed0")["t"])


def test_traj_gen_l1(golden):
    z = golden("traj_gen_l1")
    st = recipes.GenOnlyState(filled(gan.generator_shapes(4, 2)), lr=1e-4)
    lr = T("in:traj_g_lr", (4, 3, 8, 8), 0.5, 0.5)
    hr = T("in:traj_g_hr", (4, 3, 32, 32))
    ls, ps = [], []
    for _ in range(4):
        loss, out = recipes.gen_l1_step(st, lr, hr)
        ls.append(loss), ps.append(losses.psnr(out, hr))
    np.testing.assert_allclose(ls, z["loss"], rtol=1e-9)
    np.testing.assert_allclose(ps, z["psnr"], atol=1e-7)          # dB
    np.testing.assert_allclose(out.numpy(), z["y_last"], rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(st.g["conv3.weight"].detach().numpy(), z["conv3_w"], rtol=2e-6, atol=1e-7)


def test_traj_gan(golden):
    z = golden("traj_gan")
    vsd = filler.fill_state_dict({k: torch.zeros(v, dtype=DT) for k, v in vgg.vgg_shapes().items()}, salt=3)
    st = recipes.GanState(filled(gan.generator_shapes(4, 2)), filled(gan.discriminator_shapes((32, 32))), vsd,
                          lr=1e-4, vgg_resize=32, vgg_crop=28)
    lr = T("in:traj_gan_lr", (4, 3, 8, 8), 0.5, 0.5)
    hr = T("in:traj_gan_hr", (4, 3, 32, 32))
    ld, lg, ps = [], [], []
    for _ in range(3):
        a, b, fake = recipes.gan_step(st, lr, hr)
        ld.append(a), lg.append(b), ps.append(losses.psnr(fake, hr))
    np.testing.assert_allclose(ld, z["loss_d"], rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(lg, z["loss_g"], rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(ps, z["psnr"], atol=1e-7)
    np.testing.assert_allclose(fake.numpy(), z["fake_last"], rtol=2e-6, atol=1e-7)
    for k in st.g:
        if "running_" in k or "num_batches" in k:
            np.testing.assert_allclose(st.g[k].numpy(), z["g_buf/" + k], rtol=2e-6, atol=1e-7)
    for k in st.d:
        if "running_" in k or "num_batches" in k:
            np.testing.assert_allclose(st.d[k].numpy(), z["d_buf/" + k], rtol=2e-6, atol=1e-7)


def test_traj_dip(golden):
    z = golden("traj_dip")
    cfg = dip.SkipConfig(input_depth=8, skip_n33d=16, skip_n33u=16, skip_n11=4, num_scales=3)
    hr = T("in:traj_dip_hr", (1, 3, 32, 32), 0.5, 0.5)
    lr_img = downsampler.downsampler_forward(hr, 2, "lanczos2", phase=0.5, preserve_size=True)
    np.testing.assert_allclose(lr_img.numpy(), z["lr_img"], rtol=2e-6, atol=1e-7)
    zin = T("in:traj_dip_z", (1, 8, 32, 32), 0.05, 0.05)
    st = recipes.DipState(filled(dip.skip_shapes(cfg)), cfg, zin, factor=2, lr=0.01, reg_noise_std=0.05)
    ls = []
    for it in range(4):
        noise = T(f"in:traj_dip_noise{it}", (1, 8, 32, 32), 1.7)
        loss, out = recipes.dip_step(st, lr_img, noise)
        ls.append(loss)
    np.testing.assert_allclose(ls, z["loss"], rtol=1e-9)
    np.testing.assert_allclose(out.numpy(), z["out_last"], rtol=2e-6, atol=1e-7)


def test_vgg_preprocess_matches_torch_antialias():
    """transforms() restatement: pinned only against torch's own antialiased bilinear (parity unpinned
    vs torchvision, which is absent)."""
    x = T("in:vggpre", (2, 3, 40, 40)).float()
    y = vgg.preprocess(x, 32, 28)
    assert y.shape == (2, 3, 28, 28)
    x2 = T("in:vggpre2", (1, 3, 16, 16)).float()
    assert vgg.preprocess(x2, 32, 28).shape == (1, 3, 28, 28)


def test_ssim_restatement_properties():
    """oracle/metrics.ssim (PARITY UNPINNED: torchmetrics is absent and the reference holds no fixture): the published
    definition's properties -- 1 for identical images, symmetric, decreasing with added noise, window normalised."""
    from oracle import metrics
    a = T("ssim:o_a", (2, 3, 24, 30), 0.5, 0.5).float()
    n1 = T("ssim:o_n", (2, 3, 24, 30), 0.05).float()
    assert abs(float(metrics.gaussian_window().sum()) - 1.0) < 1e-12
    assert abs(metrics.ssim(a, a) - 1.0) < 1e-12
    s1, s2 = metrics.ssim(a, a + n1), metrics.ssim(a, a + 3 * n1)
    assert 0.0 < s2 < s1 < 1.0
    assert abs(metrics.ssim(a + n1, a) - s1) < 1e-12


# ----------------------------------------------------------------------------- the data-side arithmetic
def test_data_degradation_matches_reference_functions(golden):
    """oracle/data.py against what the reference's own utils/degradation.py (Pillow bicubic ``downsample``, numpy noise) and
    the direct Pillow call of dataset.py:45 produced (tests/golden/make_golden.py gen_data): all uint8, bit for bit.  The
    noise draws are repeated here from numpy's global generator with the seeds and the draw order of the reference."""
    from oracle import data as od
    z = golden("data_degradation")
    img = od.sample_image("in:data_a", 90, 124)
    d2 = od.downsample(img, 2)
    d4 = od.downsample(d2, 2)
    assert np.array_equal(d2, z["down2"]) and np.array_equal(d4, z["down4"])
    assert np.array_equal(od.downsample(img, 3), z["down3"])
    assert np.array_equal(od.resize_u8(img, 4 * d4.shape[1], 4 * d4.shape[0]), z["hr_resized"])
    assert np.array_equal(od.resize_u8(od.sample_image("in:data_b", 71, 53), 37, 50), z["odd_resized"])
    np.random.seed(7)
    noise = np.random.normal(scale=0.1 * 255, size=d2.shape)                       # utils/degradation.py:6
    assert np.array_equal(od.add_gaussian_noise(d2, noise), z["gauss"])
    np.random.seed(8)
    salt = np.random.rand(d2.shape[0], d2.shape[1]) < 0.02                        # :11
    pepper = np.random.rand(d2.shape[0], d2.shape[1]) < 0.03                      # :12
    assert np.array_equal(od.add_salt_pepper(d2, salt, pepper), z["salt_pepper"])


def test_data_resize_matches_pillow_live():
    """The same restatement against the Pillow installed beside the tests (same image here and on the GPU box): random and
    smooth images, up- and down-scaling, odd sizes."""
    PIL = pytest.importorskip("PIL.Image")
    from oracle import data as od
    rng = np.random.RandomState(0)
    for h, w, ow, oh in [(64, 96, 48, 32), (37, 53, 18, 26), (100, 77, 77, 50), (50, 60, 100, 120), (45, 80, 31, 80)]:
        img = rng.randint(0, 256, (h, w, 3), dtype=np.uint8)
        ref = np.array(PIL.fromarray(img).resize((ow, oh), PIL.BICUBIC))
        assert np.array_equal(od.resize_u8(img, ow, oh), ref), (h, w, ow, oh)


def test_data_scaling_and_patches():
    """to_tensor / scale_images / train_patch_coords against the reference's expressions evaluated literally with torch and
    numpy (dataset.py:121-159; dataset.py itself needs torchvision, which is absent here)."""
    from oracle import data as od
    img_lr = od.sample_image("in:data_lr", 24, 40)
    img_hr = od.sample_image("in:data_hr", 96, 160)
    lr_t = torch.from_numpy(img_lr).permute(2, 0, 1).contiguous().float().div(255)       # ToTensor
    hr_t = torch.from_numpy(img_hr).permute(2, 0, 1).contiguous().float().div(255)
    lr_t /= 255.0                                                                         # dataset.py:152
    hr_t /= 255.0                                                                         # :155
    hr_t *= 2                                                                             # :156
    hr_t -= 1                                                                             # :157
    lr_o, hr_o = od.scale_images(od.to_tensor(img_lr), od.to_tensor(img_hr))
    assert np.array_equal(lr_o, lr_t.numpy()) and np.array_equal(hr_o, hr_t.numpy())
    rng_a, rng_b = np.random.RandomState(5), np.random.RandomState(5)
    for _ in range(20):
        top, left, htop, hleft = od.train_patch_coords(24, 40, 16, 8, 4, rng_a)
        cx = rng_b.randint(16 // 2, 40 - 16 // 2)                                         # :128
        cy = rng_b.randint(8 // 2, 24 - 8 // 2)                                           # :129
        assert (top, left) == (int(cy - 8 // 2), int(cx - 16 // 2)) and (htop, hleft) == (top * 4, left * 4)
        assert 0 <= top and top + 8 <= 24 and 0 <= left and left + 16 <= 40


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_lowp_storage_rounds_exactly_the_stated_points(dtype):
    """oracle/lowp.py is the yardstick six GPU tests measure their bars against, so what it models is pinned here, on the CPU:
    inside lowp.storage(dtype) a convolution equals the plain fp32 convolution of the ROUNDED input and ROUNDED weight with its
    output rounded once (bias and accumulation fp32), its three gradients are the plain gradients of that computation with the
    incoming gradient, the input gradient and the weight gradient rounded; F.linear rounds input and weight (and their
    gradients) but not its output; outside the context both functions are the stock ones again."""
    import torch.nn.functional as F
    from oracle import lowp
    rd = lambda t: t.to(dtype).to(torch.float32)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 5, 9, 11, generator=g)
    w = torch.randn(7, 5, 3, 3, generator=g) * 0.2
    b = torch.randn(7, generator=g)
    go = torch.randn(2, 7, 5, 6, generator=g)
    stock_conv, stock_lin = F.conv2d, F.linear
    xs, ws, bs = (t.clone().requires_grad_(True) for t in (x, w, b))
    with lowp.storage(dtype):
        assert F.conv2d is not stock_conv and F.linear is not stock_lin
        y = F.conv2d(xs, ws, bs, stride=2, padding=1)
        y.backward(go)
    assert F.conv2d is stock_conv and F.linear is stock_lin
    xr, wr, br = (t.clone().requires_grad_(True) for t in (rd(x), rd(w), b))
    yr = F.conv2d(xr, wr, br, stride=2, padding=1)
    assert torch.equal(y.detach(), rd(yr.detach()))                   # one rounding of the output, nothing else
    assert not torch.equal(y.detach(), F.conv2d(x, w, b, stride=2, padding=1))          # (and it does change the numbers)
    yr.backward(rd(go))                                               # the gradient arriving at the output is rounded
    assert torch.equal(xs.grad, rd(xr.grad)) and torch.equal(ws.grad, rd(wr.grad))      # ... and so are the two leaving
    assert torch.equal(bs.grad, br.grad)                              # bias gradient: fp32 sum of the rounded output gradient
    # dense head: input and weight rounded, output fp32
    a = torch.randn(3, 20, generator=g)
    m = torch.randn(4, 20, generator=g)
    c = torch.randn(4, generator=g)
    gl = torch.randn(3, 4, generator=g)
    a_s, m_s = a.clone().requires_grad_(True), m.clone().requires_grad_(True)
    with lowp.storage(dtype):
        z = F.linear(a_s, m_s, c)
        z.backward(gl)
    ar, mr = rd(a).requires_grad_(True), rd(m).requires_grad_(True)
    zr = F.linear(ar, mr, c)
    assert torch.equal(z.detach(), zr.detach())                       # not rounded
    zr.backward(gl)
    assert torch.equal(a_s.grad, rd(ar.grad)) and torch.equal(m_s.grad, rd(mr.grad))
    # an exception inside the block restores the stock functions too
    with pytest.raises(RuntimeError):
        with lowp.storage(dtype):
            raise RuntimeError("x")
    assert F.conv2d is stock_conv and F.linear is stock_lin
