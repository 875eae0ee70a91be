"""GPU: discriminator, VGG perceptual loss, DIP skip net, downsampler and the GAN / DIP step recipes on the HIP
path against the CPU oracle (fp32) on identical parameters and inputs.  bf16 tolerances are stated inline."""
import importlib

import numpy as np
import pytest
import torch
import torch.nn.functional as TF

from oracle import dip, downsampler, filler, gan, losses, lowp, recipes, vgg

pytestmark = pytest.mark.gpu
PKG = "deep-super-resolution_amd"


def P(sub):
    return importlib.import_module(PKG + "." + sub)


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    P("_lib").lib()
    return torch.device("cuda:0")


def bfr(t):
    return t.to(torch.bfloat16).float()


def rel_err(got, ref):
    got, ref = got.double(), ref.double()
    return float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-20))


def cos(a, b):
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    return float((a @ b) / (a.norm() * b.norm()).clamp_min(1e-30))


def to_nhwc(x):
    n, c, h, w = x.shape
    cp = (c + 7) // 8 * 8
    out = torch.zeros(n, h, w, cp, dtype=torch.bfloat16)
    out[..., :c] = x.permute(0, 2, 3, 1).to(torch.bfloat16)
    return out


def from_nhwc(y, c):
    return y[..., :c].float().permute(0, 3, 1, 2).contiguous()


# ----------------------------------------------------------------------------- single ops
@pytest.mark.parametrize("n", [5, 40])
def test_dense_head(dev, n):
    F = P("functional")
    c, h, w = 512, 2, 3
    k = c * h * w
    x = bfr(filler.tensor("dh:x", (n, c, h, w)))
    w1 = bfr(filler.tensor("dh:w1", (1024, k), float(np.sqrt(3.0 / k))))
    b1 = filler.tensor("dh:b1", (1024,), 0.1)
    w2 = filler.tensor("dh:w2", (1, 1024), float(np.sqrt(3.0 / 1024)))
    b2 = filler.tensor("dh:b2", (1,), 0.1)
    xr, w1r, b1r, w2r, b2r = [t.clone().requires_grad_(True) for t in (x, w1, b1, w2, b2)]
    hr = TF.leaky_relu(TF.linear(xr.reshape(n, -1), w1r, b1r), 0.2)
    outr = torch.sigmoid(TF.linear(hr, w2r, b2r))
    probe = filler.tensor("dh:p", (n, 1))
    (outr * probe).sum().backward()
    xg = to_nhwc(x).to(dev).requires_grad_(True)
    gs = [t.to(dev).requires_grad_(True) for t in (w1, b1, w2, b2)]
    out = F.DenseHead.apply(xg, *gs, c)
    out.backward(probe.to(dev))
    torch.cuda.synchronize()
    assert rel_err(out.detach().cpu(), outr.detach()) < 5e-3
    assert rel_err(from_nhwc(xg.grad.cpu(), c), xr.grad) < 2e-2
    for got, ref, name in zip(gs, (w1r, b1r, w2r, b2r), ("w1", "b1", "w2", "b2")):
        assert rel_err(got.grad.cpu(), ref.grad) < 2e-2, name


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("b,k,o", [(37, 1000, 72), (64, 4104, 1024), (3, 264, 8), (32, 33000, 1032)])
def test_linear_kernels_ragged(dev, dtype, b, k, o):
    """dsr_linear_{fwd,dgrad,wgrad} through the C ABI at sizes that are not multiples of any tile (dense1 of
    discriminator.py:41 is 64 x 524288 x 1024; these shapes exercise every edge guard) against fp64 matmuls."""
    import ctypes as C
    L = P("_lib")
    lib = L.lib()
    dt = 0 if dtype == torch.bfloat16 else 1
    ptr = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    x = filler.tensor("lin:x", (b, k)).to(dtype)
    w = filler.tensor("lin:w", (o, k), float(np.sqrt(3.0 / k))).to(dtype)
    dy = filler.tensor("lin:dy", (b, o)).to(dtype)
    bias = filler.tensor("lin:b", (o,), 0.1)
    xd, wd, dyd, bd = x.to(dev), w.to(dev), dy.to(dev), bias.to(dev)
    # forward
    wsz = lib.dsr_linear_fwd_workspace(b, k, o)
    ws = torch.empty(max(wsz, 16), dtype=torch.uint8, device=dev)
    out = torch.empty((b, o), dtype=torch.float32, device=dev)
    L.check(lib.dsr_linear_fwd(dt, ptr(xd), ptr(wd), ptr(bd), 1, 0.2, ptr(out), b, k, o, ptr(ws), wsz, st))
    ref = TF.leaky_relu(x.double() @ w.double().t() + bias.double(), 0.2)
    assert (out.cpu().double() - ref).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item())
    # dgrad
    dx = torch.full((b, k), float("nan"), dtype=dtype, device=dev)
    L.check(lib.dsr_linear_dgrad(dt, ptr(dyd), ptr(wd), ptr(dx), b, o, k, st))
    ref = dy.double() @ w.double()
    tol = (2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11) * ref.abs().max().item() + 1e-6
    assert (dx.cpu().double() - ref).abs().max().item() <= tol
    # wgrad (batch-minor operands, batch padded to 32 | 64)
    bp = 32 if b <= 32 else 64
    dyt = torch.zeros((o, bp), dtype=dtype, device=dev)
    dyt[:, :b] = dyd.t()
    xt = torch.zeros((k, bp), dtype=dtype, device=dev)
    xt[:, :b] = xd.t()
    dw = torch.full((o, k), float("nan"), dtype=torch.float32, device=dev)
    L.check(lib.dsr_linear_wgrad(dt, ptr(dyt), ptr(xt), ptr(dw), bp, o, k, st))
    ref = dy.double().t() @ x.double()
    assert (dw.cpu().double() - ref).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item())


def test_linear_wgrad_gathered(dev):
    """dsr_linear_wgrad_gathered (data-parallel dense head: summed gradient from all-gathered rank-local factors) equals
    the mean of the per-rank dsr_linear_wgrad results."""
    import ctypes as C
    L = P("_lib")
    lib = L.lib()
    ptr = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    R, o, k, bp = 3, 200, 1160, 64
    dyt = (torch.rand(R, o, bp, device=dev) - 0.5).to(torch.bfloat16)
    xt = (torch.rand(R, k, bp, device=dev) - 0.5).to(torch.bfloat16)
    dw = torch.full((o, k), float("nan"), dtype=torch.float32, device=dev)
    L.check(lib.dsr_linear_wgrad_gathered(0, ptr(dyt), ptr(xt), ptr(dw), bp, o, k, R, 1.0 / R, st))
    ref = torch.zeros((o, k), dtype=torch.float64, device=dev)
    for r in range(R):
        one = torch.empty((o, k), dtype=torch.float32, device=dev)
        L.check(lib.dsr_linear_wgrad(0, ptr(dyt[r]), ptr(xt[r]), ptr(one), bp, o, k, st))
        ref += one.double()
    ref /= R
    torch.cuda.synchronize()
    assert (dw.double() - ref).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item())
    exact = (dyt.double().permute(1, 0, 2).reshape(o, R * bp) @ xt.double().permute(1, 0, 2).reshape(k, R * bp).t()) / R
    assert (dw.double() - exact).abs().max().item() <= 1e-5 * max(1.0, exact.abs().max().item())


def test_maxpool_bilinear_concat(dev):
    F = P("functional")
    x = bfr(filler.tensor("mp:x", (2, 16, 6, 10)))
    xr = x.clone().requires_grad_(True)
    yr = TF.max_pool2d(xr, 2, 2)
    probe = bfr(filler.tensor("mp:p", tuple(yr.shape)))
    (yr * probe).sum().backward()
    xg = to_nhwc(x).to(dev).requires_grad_(True)
    yg = F.MaxPool2.apply(xg)
    yg.backward(to_nhwc(probe).to(dev))
    assert torch.equal(from_nhwc(yg.detach().cpu(), 16), yr.detach())
    assert torch.equal(from_nhwc(xg.grad.cpu(), 16), xr.grad)
    # bilinear x2
    x = bfr(filler.tensor("bl:x", (2, 8, 5, 7)))
    xr = x.clone().requires_grad_(True)
    yr = TF.interpolate(xr, scale_factor=2, mode="bilinear", align_corners=False)
    probe = bfr(filler.tensor("bl:p", tuple(yr.shape)))
    (yr * probe).sum().backward()
    xg = to_nhwc(x).to(dev).requires_grad_(True)
    yg = F.Bilinear2x.apply(xg)
    yg.backward(to_nhwc(probe).to(dev))
    assert rel_err(from_nhwc(yg.detach().cpu(), 8), yr.detach()) < 5e-3
    assert rel_err(from_nhwc(xg.grad.cpu(), 8), xr.grad) < 5e-3
    # nearest x2 (a copy forward, a 4-term fp32 sum rounded once backward)
    x = bfr(filler.tensor("nn:x", (2, 11, 5, 7)))
    xr = x.clone().requires_grad_(True)
    yr = TF.interpolate(xr, scale_factor=2, mode="nearest")
    probe = bfr(filler.tensor("nn:p", tuple(yr.shape)))
    (yr * probe).sum().backward()
    xg = to_nhwc(x).to(dev).requires_grad_(True)
    yg = F.Nearest2x.apply(xg)
    yg.backward(to_nhwc(probe).to(dev))
    assert torch.equal(from_nhwc(yg.detach().cpu(), 11), yr.detach())
    assert rel_err(from_nhwc(xg.grad.cpu(), 11), xr.grad) < 4e-3       # one bf16 rounding of an fp32 sum
    # average pool 2x2, odd sizes (floor mode drops the trailing row / column; their gradient is zero)
    x = bfr(filler.tensor("ap:x", (2, 9, 7, 10)))
    xr = x.clone().requires_grad_(True)
    yr = TF.avg_pool2d(xr, 2, 2)
    probe = bfr(filler.tensor("ap:p", tuple(yr.shape)))
    (yr * probe).sum().backward()
    xg = to_nhwc(x).to(dev).requires_grad_(True)
    yg = F.AvgPool2.apply(xg)
    assert tuple(yg.shape) == (2, 3, 5, 16)
    yg.backward(to_nhwc(probe).to(dev))
    assert rel_err(from_nhwc(yg.detach().cpu(), 9), yr.detach()) < 4e-3
    assert torch.equal(from_nhwc(xg.grad.cpu(), 9), bfr(xr.grad))      # dy / 4 is exact in bf16
    assert float(xg.grad[:, 6].float().abs().max()) == 0.0
    # concat with centre crop, channel counts that are not multiples of 8
    a = bfr(filler.tensor("cc:a", (1, 4, 10, 12)))
    b = bfr(filler.tensor("cc:b", (1, 13, 8, 8)))
    ar, br = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = dip.concat_center_crop([ar, br])
    probe = bfr(filler.tensor("cc:p", tuple(yr.shape)))
    (yr * probe).sum().backward()
    ag, bg = to_nhwc(a).to(dev).requires_grad_(True), to_nhwc(b).to(dev).requires_grad_(True)
    yg = F.ConcatCrop.apply(ag, bg, 4, 13)
    assert tuple(yg.shape) == (1, 8, 8, 24)
    yg.backward(to_nhwc(probe).to(dev))
    assert torch.equal(from_nhwc(yg.detach().cpu(), 17), yr.detach())
    assert float(yg.detach()[..., 17:].float().abs().max()) == 0.0
    assert torch.equal(from_nhwc(ag.grad.cpu(), 4), ar.grad) and torch.equal(from_nhwc(bg.grad.cpu(), 13), br.grad)


def test_bn_act_standalone(dev):
    F = P("functional")
    c = 132
    x = bfr(filler.tensor("bna:x", (2, c, 6, 6)))
    gamma, beta = filler.tensor("bna:g", (c,), 0.2, 1.0), filler.tensor("bna:b", (c,), 0.1)
    xr, gr, br = x.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm, rv = torch.zeros(c), torch.ones(c)
    yr = TF.batch_norm(xr, rm, rv, gr, br, training=True, momentum=0.1, eps=1e-5)
    probe = bfr(filler.tensor("bna:p", tuple(yr.shape)))
    (yr * probe).sum().backward()
    xg = to_nhwc(x).to(dev).requires_grad_(True)
    gg, bg = gamma.to(dev).requires_grad_(True), beta.to(dev).requires_grad_(True)
    rmg, rvg, nbt = torch.zeros(c, device=dev), torch.ones(c, device=dev), torch.zeros((), dtype=torch.int64, device=dev)
    yg = F.BNAct.apply(xg, gg, bg, rmg, rvg, nbt, c, dict(act=F.ACT_NONE, train=True))
    yg.backward(to_nhwc(probe).to(dev))
    assert rel_err(from_nhwc(yg.detach().cpu(), c), yr.detach()) < 1e-2
    assert rel_err(from_nhwc(xg.grad.cpu(), c), xr.grad) < 3e-2
    assert rel_err(gg.grad.cpu(), gr.grad) < 2e-2 and rel_err(bg.grad.cpu(), br.grad) < 2e-2
    assert rel_err(rmg.cpu(), rm) < 1e-3 and rel_err(rvg.cpu(), rv) < 1e-3


@pytest.mark.parametrize("shape,resize,crop", [((2, 3, 40, 40), 32, 28), ((1, 3, 16, 24), 32, 28), ((1, 3, 64, 64), 256, 224)])
def test_resize_norm(dev, shape, resize, crop):
    F, G = P("functional"), P("utils.GAN")
    x = filler.tensor("rn:x" + str(shape), shape)
    xr = x.clone().requires_grad_(True)
    yr = vgg.preprocess(xr, resize, crop)
    probe = bfr(filler.tensor("rn:p" + str(shape), tuple(yr.shape)))
    (yr * probe).sum().backward()
    tab = G.ResampleTables(shape[2], shape[3], dev, resize, crop)
    xg = x.to(dev).requires_grad_(True)
    yg = F.ResizeNorm.apply(xg, tab, torch.bfloat16)
    assert tuple(yg.shape) == (shape[0], crop, crop, 8)
    yg.backward(to_nhwc(probe).to(dev))
    assert rel_err(from_nhwc(yg.detach().cpu(), 3), yr.detach()) < 6e-3
    assert rel_err(xg.grad.cpu(), xr.grad) < 1e-4


def test_downsampler_module(dev, golden):
    D = P("utils.downsampler")
    z = golden("downsampler")
    for f in (2, 4, 8):
        d = D.Downsampler(3, f, "lanczos2", phase=0.5, preserve_size=True).to(dev)
        np.testing.assert_allclose(d.kernel, z[f"kernel_f{f}"], rtol=1e-12, atol=1e-15)
        assert tuple(d.downsampler_.weight.shape) == (3, 3, 4 * f, 4 * f)
        x = filler.tensor(f"in:down{f}", (1, 3, 32, 32), 0.5, 0.5).to(dev).requires_grad_(True)
        y = d(x)
        probe = filler.tensor(f"probe:down{f}", tuple(y.shape))
        (y * probe.to(dev)).sum().backward()
        np.testing.assert_allclose(y.detach().cpu().numpy(), z[f"y_f{f}"], rtol=2e-5, atol=2e-6)      # fp32 kernel
        np.testing.assert_allclose(x.grad.cpu().numpy(), z[f"gx_f{f}"], rtol=2e-5, atol=2e-6)
    x = filler.tensor("in:downl3", (2, 3, 16, 20), 0.5, 0.5).to(dev)
    for tag, args in (("l3", dict(factor=2, kernel_type="lanczos3", phase=0, preserve_size=True)),
                      ("g12", dict(factor=2, kernel_type="gauss12", phase=0, preserve_size=True)),
                      ("box", dict(factor=4, kernel_type="box", phase=0.5, kernel_width=4, preserve_size=False))):
        d = D.Downsampler(3, **args).to(dev)
        np.testing.assert_allclose(d(x).cpu().numpy(), z["y_" + tag], rtol=2e-5, atol=2e-6)
    with pytest.raises(AssertionError):
        D.Downsampler(3, 2, "nope")


# ----------------------------------------------------------------------------- modules
def grads_ok(module, osd, min_cos=0.97, ratio=0.12):
    bad = []
    for k, p in module.named_parameters():
        if p.grad is None:
            continue
        ref = osd[k].grad
        if ref is None or ref.abs().sum() < 1e-3 * max(1.0, ref.numel() ** 0.5) or ref.numel() == 1:
            continue
        c = cos(p.grad.cpu(), ref)
        r = float(p.grad.norm().cpu() / ref.norm())
        if c < min_cos or abs(r - 1) > ratio:
            bad.append((k, round(c, 4), round(r, 4)))
    return bad


@pytest.mark.parametrize("need_dx", [False, True], ids=["image_input", "input_needs_grad"])
def test_discriminator_fused_first_two_layers_equal_two_launches(dev, need_dx, monkeypatch):
    """Discriminator.features runs its first two convolutions as ONE kernel in train mode (functional ConvAct defer +
    ConvBNAct first2 -> dsr_conv_first2_fwd); DSR_CONV_FIRST2=0 runs the two launches.  Same module, same weights, same batch:
    output, every parameter gradient, the input gradient (when the image requires grad: the first layer's activation must then
    have been written for the unfused backward path), BatchNorm running statistics -- and the no-grad pass, which never writes
    the first layer's activation."""
    Dm = P("models.GAN.discriminator")
    hw = (64, 96)
    sd = filler.fill_state_dict(gan.template(gan.discriminator_shapes(hw)))
    x = filler.tensor("in:disc_f2", (3, 3, hw[0], hw[1]))
    probe = filler.tensor("probe:disc_f2", (3, 1)).to(dev)
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("DSR_CONV_FIRST2", mode)
        d = Dm.Discriminator(hw)
        d.load_state_dict(sd)
        d.to(dev).train()
        xg = x.to(dev).requires_grad_(need_dx)
        y = d(xg)
        (y * probe).sum().backward()
        with torch.no_grad():
            y2 = d(x.to(dev))
        torch.cuda.synchronize()
        res[mode] = (y.detach().clone(), y2.clone(), {k: p.grad.clone() for k, p in d.named_parameters() if p.grad is not None},
                     xg.grad.clone() if need_dx else None, {k: v.clone() for k, v in d.state_dict().items() if "running_" in k})
    ya, y2a, ga, dxa, ra = res["1"]
    yb, y2b, gb, dxb, rb = res["0"]
    assert torch.isfinite(ya).all() and (ya - yb).abs().max().item() < 2e-3 and (y2a - y2b).abs().max().item() < 2e-3
    assert set(ga) == set(gb)
    # (the fused kernel adds the second layer's bias before the products instead of after: a last-bit difference in some bf16
    #  outputs, which seven train-mode BatchNorms at batch 3 amplify towards the first layers -- the same mechanism as the
    #  storage floor of tests/parity_util.py; measured: cosine 0.9976 on conv.weight, > 0.999 from the third block on)
    for k in ga:
        assert cos(ga[k], gb[k]) > 0.99 and abs(float(ga[k].norm() / gb[k].norm()) - 1) < 3e-2, k
    if need_dx:
        assert cos(dxa, dxb) > 0.99
    for k in ra:
        assert rel_err(ra[k], rb[k]) < 2e-3, k


def test_discriminator_batchnorm_sums_in_the_next_layers_input_gradient(dev, monkeypatch):
    """A stride-2 block's input-gradient launch also forms the BatchNorm-backward sums of the block in front of it
    (dsr_conv_dgrad_bn through the link Discriminator.features sets up); functional.DGRAD_BN off runs the separate reduce pass.
    Same module, weights and batch at 128 x 128 (blocks 1 and 3 qualify: tiles of whole rows of one image; block 5's 8 x 8
    gradient map does not and keeps its reduce pass): identical output, the same launches taken as expected, every gradient
    equal to fp32 summation order."""
    Dm = P("models.GAN.discriminator")
    F = P("functional")
    hw = (128, 128)
    n = 2
    sd = filler.fill_state_dict(gan.template(gan.discriminator_shapes(hw)))
    x = filler.tensor("in:disc_bn", (n, 3, hw[0], hw[1]))
    probe = filler.tensor("probe:disc_bn", (n, 1)).to(dev)
    res = {}
    for on in (True, False):
        monkeypatch.setattr(F, "DGRAD_BN", on)
        d = Dm.Discriminator(hw)
        d.load_state_dict(sd)
        d.to(dev).train()
        xg = x.to(dev).requires_grad_(True)
        F.KERNEL_LOG = []
        try:
            y = d(xg)
            (y * probe).sum().backward()
            torch.cuda.synchronize()
            names = [e[4] for e in F.KERNEL_LOG]
        finally:
            F.KERNEL_LOG = None
        assert sum("kernel<bn>" in nm for nm in names) == (2 if on else 0), names
        res[on] = (y.detach().clone(), {k: p.grad.clone() for k, p in d.named_parameters() if p.grad is not None}, xg.grad.clone())
    (ya, ga, dxa), (yb, gb, dxb) = res[True], res[False]
    assert torch.equal(ya, yb) and set(ga) == set(gb)
    for k in ga:
        assert torch.isfinite(ga[k]).all() and rel_err(ga[k], gb[k]) < 2e-3, (k, rel_err(ga[k], gb[k]))
    # (the image gradient passes through bf16 tensors: a last-bit change in a BatchNorm coefficient flips single roundings, one
    #  bf16 ulp = 0.4 % of an element)
    assert rel_err(dxa, dxb) < 1e-2 and cos(dxa, dxb) > 0.99999


def test_discriminator_fused_backward_of_first_two_layers(dev, monkeypatch):
    """An image that needs no gradient (the discriminator's own update, train_GAN.py:47-56): ConvBNAct.backward of the second
    layer runs its input gradient AND the image layer's whole backward as one launch (dsr_conv_dgrad_first_bwd) and hands the
    image layer's gradients to its autograd node through the link; DSR_FIRST2_BACKWARD off runs the separate launches.  Same
    module, weights and batch: the output and every gradient except the image layer's are BIT-identical (nothing else
    changed), conv.weight / conv.bias agree to the rounding of the masked gradient (once vs twice) and with the oracle.  Needs
    an image row of 512 pixels (a tile is 256 gradient pixels of ONE row); with an image that requires grad the launch is
    not taken and the two settings are bit-identical throughout."""
    Dm = P("models.GAN.discriminator")
    F = P("functional")
    hw = (16, 512)
    n = 2
    sd = filler.fill_state_dict(gan.template(gan.discriminator_shapes(hw)))
    x = filler.tensor("in:disc_fb", (n, 3, hw[0], hw[1]))
    probe = filler.tensor("probe:disc_fb", (n, 1)).to(dev)
    res = {}
    names = []
    for need_dx in (False, True):
        for on in (True, False):
            monkeypatch.setattr(F, "FIRST2_BACKWARD", on)
            d = Dm.Discriminator(hw)
            d.load_state_dict(sd)
            d.to(dev).train()
            xg = x.to(dev).requires_grad_(need_dx)
            F.KERNEL_LOG = []
            try:
                y = d(xg)
                (y * probe).sum().backward()
                torch.cuda.synchronize()
                names = [e[4] for e in F.KERNEL_LOG]
            finally:
                F.KERNEL_LOG = None
            took = any("first_bwd>" in nm for nm in names)
            assert took == (on and not need_dx), (need_dx, on, names)
            res[(need_dx, on)] = (y.detach().clone(), {k: p.grad.clone() for k, p in d.named_parameters() if p.grad is not None})
    for need_dx in (False, True):
        (ya, ga), (yb, gb) = res[(need_dx, True)], res[(need_dx, False)]
        assert torch.equal(ya, yb) and set(ga) == set(gb) and "conv.weight" in ga and "conv.bias" in ga
        for k in ga:
            if need_dx or k not in ("conv.weight", "conv.bias"):
                assert torch.equal(ga[k], gb[k]), (need_dx, k)
            else:
                assert torch.isfinite(ga[k]).all() and rel_err(ga[k], gb[k]) < 5e-3, (k, rel_err(ga[k], gb[k]))
    # against the oracle (fp32 CPU restatement of the reference): the first layer's gradients of the fused launch
    osd = {k: v.clone() for k, v in sd.items()}
    recipes.leaves(osd)
    yr = gan.discriminator_forward(osd, x, True)
    (yr * probe.cpu()).sum().backward()
    ga, gb = res[(False, True)][1], res[(False, False)][1]
    for k in ("conv.weight", "conv.bias"):      # (batch 2 through seven train-mode BatchNorms: the bf16 floor of parity_util, not 1.0)
        cf, cu = cos(ga[k].cpu(), osd[k].grad), cos(gb[k].cpu(), osd[k].grad)
        assert cf > 0.9 and cf > cu - 0.01, (k, cf, cu)


@pytest.mark.parametrize("hw,n", [((32, 32), 4), ((48, 32), 3), ((64, 64), 4)])
def test_discriminator(dev, hw, n):
    Dm = P("models.GAN.discriminator")
    sd = filler.fill_state_dict(gan.template(gan.discriminator_shapes(hw)))
    d = Dm.Discriminator(hw)
    assert list(d.state_dict().keys()) == list(sd.keys())
    assert all(tuple(d.state_dict()[k].shape) == tuple(sd[k].shape) for k in sd)
    d.load_state_dict(sd)
    d.to(dev).train()
    x = filler.tensor("in:disc" + str(hw), (n, 3, hw[0], hw[1]))
    xg = x.to(dev).requires_grad_(True)
    y = d(xg)
    assert tuple(y.shape) == (n, 1) and y.dtype == torch.float32
    probe = filler.tensor("probe:disc" + str(hw), (n, 1))
    (y * probe.to(dev)).sum().backward()
    osd = {k: v.clone() for k, v in sd.items()}
    recipes.leaves(osd)
    xr = x.clone().requires_grad_(True)
    yr = gan.discriminator_forward(osd, xr, True)
    (yr * probe).sum().backward()
    assert (y.detach().cpu() - yr.detach()).abs().max().item() < 0.03          # probabilities in (0,1)
    assert cos(xg.grad.cpu(), xr.grad) > 0.97
    assert not grads_ok(d, osd), grads_ok(d, osd)
    for k, v in d.state_dict().items():
        if "running_" in k:
            assert rel_err(v.cpu(), osd[k]) < 2e-2, k
    # forward_pair == two separate calls (same outputs, same BatchNorm bookkeeping)
    d2 = Dm.Discriminator(hw)
    d2.load_state_dict(sd)
    d2.to(dev).train()
    x2 = filler.tensor("in:disc2" + str(hw), (n, 3, hw[0], hw[1]))
    with torch.no_grad():
        pa, pb = d2.forward_pair(x.to(dev), x2.to(dev))
        d3 = Dm.Discriminator(hw)
        d3.load_state_dict(sd)
        d3.to(dev).train()
        qa, qb = d3(x.to(dev)), d3(x2.to(dev))
    assert (pa - qa).abs().max().item() < 2e-3 and (pb - qb).abs().max().item() < 2e-3
    for k, v in d2.state_dict().items():
        assert torch.equal(v, d3.state_dict()[k]) or "dense" in k or v.dtype != torch.float32 or \
            (v - d3.state_dict()[k]).abs().max().item() < 1e-6, k


def test_vgg_loss(dev):
    G = P("utils.GAN")
    m = G.Vgg19Loss(resize_to=48, crop=40).to(dev)
    keys = list(m.state_dict().keys())
    assert keys[0] == "net.0.0.weight" and keys[-1] == "net.0.34.bias" and len(keys) == 32
    osd = {k[len("net.0."):]: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    a = filler.tensor("vgg:a", (2, 3, 64, 64))
    b = filler.tensor("vgg:b", (2, 3, 64, 64))
    ar = a.clone().requires_grad_(True)
    lref = vgg.vgg_loss(osd, ar, b, 48, 40)
    lref.backward()
    ag = a.to(dev).requires_grad_(True)
    l = m(ag, b.to(dev))
    l.backward()
    an = a.clone().requires_grad_(True)
    with lowp.storage(torch.bfloat16):                    # 16-bit-storage floor of the same computation
        vgg.vgg_loss(osd, an, b, 48, 40).backward()
    floor = 1 - cos(an.grad, ar.grad)
    assert abs(l.item() - lref.item()) < 3e-2 * abs(lref.item()), (l.item(), lref.item())
    assert 1 - cos(ag.grad.cpu(), ar.grad) <= 3.0 * floor + 0.01, (cos(ag.grad.cpu(), ar.grad), floor)
    assert abs(float(ag.grad.norm().cpu() / ar.grad.norm()) - 1) < 0.1


DIP_CASES = [("small", (1, 8, 32, 32), dict(skip_n33d=16, skip_n33u=16, skip_n11=4, num_scales=3)),
             ("full64", (1, 32, 64, 64), {}), ("crop72x104", (1, 32, 72, 104), {})]
_S3 = dict(skip_n33d=16, skip_n33u=16, skip_n11=4, num_scales=3)
# get_net's optional arguments (models/DIP/__init__.py:8): act_fun, downsample_mode, upsample_mode
DIP_CASES += [
    ("opt_elu_avg_nearest", (1, 8, 32, 32), dict(_S3, act_fun="ELU", downsample_mode="avg", upsample_mode="nearest")),
    ("opt_none_max_bilinear", (1, 8, 32, 32), dict(_S3, act_fun="none", downsample_mode="max", upsample_mode="bilinear")),
    ("opt_leaky_avg_nearest_odd", (1, 8, 36, 44), dict(_S3, act_fun="LeakyReLU", downsample_mode="avg",
                                                       upsample_mode="nearest")),
    ("opt_elu_stride_nearest_b2", (2, 8, 32, 48), dict(_S3, act_fun="ELU", downsample_mode="stride",
                                                       upsample_mode="nearest")),
]


@pytest.mark.parametrize("tag,shape,kw", DIP_CASES)
def test_dip_skip_net(dev, tag, shape, kw):
    M = P("models.DIP")
    cfg = dip.SkipConfig(input_depth=shape[1], **kw)
    sd = filler.fill_state_dict(gan.template(dip.skip_shapes(cfg)))
    nkw = dict(kw)
    net = M.get_net(shape[1], "skip", "reflection", upsample_mode=nkw.pop("upsample_mode", "bilinear"), **nkw)
    assert set(net.state_dict().keys()) == set(sd.keys())
    net.load_state_dict(sd)
    net.to(dev).train()
    x = filler.tensor("in:dipg_" + tag, shape, 0.05, 0.05)
    xg = x.to(dev).requires_grad_(True)
    y = net(xg)
    assert y.dtype == torch.float32
    if "odd" not in tag:      # sizes that do not halve evenly are centre-cropped by Concat: shape checked against the oracle
        assert tuple(y.shape) == (shape[0], 3, shape[2], shape[3])
    probe = filler.tensor("probe:dipg_" + tag, tuple(y.shape))
    (y * probe.to(dev)).sum().backward()
    osd = {k: v.clone() for k, v in sd.items()}
    recipes.leaves(osd)
    xr = x.clone().requires_grad_(True)
    yr = dip.skip_forward(osd, xr, cfg, True)
    assert tuple(yr.shape) == tuple(y.shape)
    (yr * probe).sum().backward()
    # fp16-storage floor of the same computation (oracle/lowp.py); SkipNet computes in fp16 by default
    assert net.compute_dtype == torch.float16
    nsd = {k: v.clone() for k, v in sd.items()}
    recipes.leaves(nsd)
    with lowp.storage(torch.float16):
        yn = dip.skip_forward(nsd, x.clone(), cfg, True)
        (yn * probe).sum().backward()
    err = (y.detach().cpu() - yr.detach()).abs().max().item()
    ferr = (yn.detach() - yr.detach()).abs().max().item()
    assert err <= 3.0 * ferr + 0.01, (err, ferr)                   # sigmoid output in (0,1)
    bad = []
    for k, p in net.named_parameters():
        ref = osd[k].grad
        if ref.abs().sum() < 1e-3 * max(1.0, ref.numel() ** 0.5):
            continue
        if p.grad is None:        # conv bias in front of a train-mode BatchNorm: analytically zero, not produced (the oracle
            assert k.endswith(".bias") and osd[k].dim() == 1      # holds cancellation noise there, large at 2x2 populations)
            continue
        c, cf = cos(p.grad.cpu(), ref), cos(nsd[k].grad, ref)
        if (1 - c) > 3.0 * (1 - cf) + 0.02:
            bad.append((k, round(c, 4), round(cf, 4)))
    # The innermost scale normalises 2x2 ... 8x8 maps at batch 1 (4 ... 64 values per channel): gradients of THAT scale's
    # parameters are chaotic under ANY 16-bit storage (two fp16 implementations differ there as much as each differs from
    # fp32).  Only tensors of the innermost scale may miss the bound, and even those must stay correlated.
    inner = "1.1.7." * (len(cfg.down) - 1)            # key prefix of the deepest scale (models/DIP/skip.py nesting)
    _record_dip(tag, bad)
    assert not bad, (inner, bad)      # measured round 2: no tensor misses the bound in any of the seven configurations


def _record_dip(tag, bad):
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_dip_skip.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        data = json.load(open(path)) if os.path.exists(path) else {}
        data[tag] = bad
        json.dump(data, open(path, "w"), indent=1)
    except OSError:
        pass


def _torch_forward(m, x):
    """Plain torch (CPU fp32) execution of a module tree built by skip(): the leaves are stock nn modules; Concat is the
    reference's centre-crop concatenation (models/DIP/utils.py:18-38)."""
    import torch.nn as nn
    DU = P("models.DIP.utils")
    if isinstance(m, DU.Concat):
        return dip.concat_center_crop([_torch_forward(b, x) for b in m.children()])
    if isinstance(m, nn.Sequential):
        for c in m.children():
            x = _torch_forward(c, x)
        return x
    return m(x)


@pytest.mark.parametrize("tag,opts", [
    ("nobias_zero", dict(need_bias=False, pad="zero", upsample_mode="bilinear")),
    ("nosigmoid_no1x1", dict(need_sigmoid=False, need1x1_up=False, pad="reflection", upsample_mode="nearest")),
    ("noskip_level", dict(num_channels_skip=[4, 0, 4], pad="reflection", upsample_mode="bilinear")),
])
def test_dip_skip_builder_flags(dev, tag, opts):
    """skip()'s remaining flags (need_bias, need_sigmoid, need1x1_up, zero padding, a scale without skip branch): the HIP
    executor against the SAME module tree run by plain torch on the CPU in fp32 (the leaves are stock nn modules)."""
    import copy
    S = P("models.DIP.skip")
    torch.manual_seed(11)
    kw = dict(num_channels_down=[16, 16, 16], num_channels_up=[16, 16, 16], num_channels_skip=[4, 4, 4])
    kw.update(opts)
    net = S.skip(8, 3, **kw)
    for mod in net.modules():                       # non-trivial BatchNorm affine parameters
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.weight.data.uniform_(0.5, 1.5)
            mod.bias.data.uniform_(-0.3, 0.3)
    ref = copy.deepcopy(net).float().train()
    low = copy.deepcopy(net).float().train()        # the same tree once more, for the fp16-storage floor (oracle/lowp.py)
    net.to(dev).train()
    x = filler.tensor("in:dipflags_" + tag, (1, 8, 32, 32), 0.05, 0.05)
    xr = x.clone().requires_grad_(True)
    yr = _torch_forward(ref, xr)
    probe = filler.tensor("probe:dipflags_" + tag, tuple(yr.shape))
    (yr * probe).sum().backward()
    with lowp.storage(torch.float16):
        (_torch_forward(low, x.clone()) * probe).sum().backward()
    xg = x.to(dev).requires_grad_(True)
    y = net(xg)
    assert y.dtype == torch.float32 and tuple(y.shape) == tuple(yr.shape)
    (y * probe.to(dev)).sum().backward()
    scale = float(yr.detach().abs().max())
    assert float((y.detach().cpu() - yr.detach()).abs().max()) <= 0.03 * max(scale, 1.0), tag
    assert cos(xg.grad.cpu(), xr.grad) >= 0.97, cos(xg.grad.cpu(), xr.grad)
    refp, lowp_ = dict(ref.named_parameters()), dict(low.named_parameters())
    bad = []
    for k, p_ in net.named_parameters():
        r = refp[k].grad
        if r is None or r.abs().sum() < 1e-3 * max(1.0, r.numel() ** 0.5) or r.numel() < 16:
            continue
        # every tensor by the floor rule of test_dip_skip_net (no unnamed exceptions): the HIP gradient may sit at most three
        # times as far from the fp32 one as the fp16-storage restatement of the same tree does, + 0.02
        c, cf = cos(p_.grad.cpu(), r), cos(lowp_[k].grad, r)
        if (1 - c) > 3.0 * (1 - cf) + 0.02:
            bad.append((k, round(c, 4), round(cf, 4)))
    _record_dip("flags_" + tag, bad)
    assert not bad, bad


# ----------------------------------------------------------------------------- step recipes
@pytest.mark.parametrize("overlap", [True, False])
def test_gan_step_vs_oracle(dev, overlap):
    """train_GAN.py:38-71 for 2 steps on small shapes; stand-in VGG (resize 32 / crop 28); two-stream and single-stream
    form of the step.  |dPSNR| <= 0.02 dB (the north-star bar) and losses within 2 % at both steps; after the steps the
    BatchNorm running statistics of both networks and every weight tensor's 2-step displacement agree with the oracle."""
    Gm, Dm, GANu, optim, steps = P("models.GAN.generator"), P("models.GAN.discriminator"), P("utils.GAN"), P("optim"), P("steps")
    gsd = filler.fill_state_dict(gan.template(gan.generator_shapes(4, 2)))
    dsd = filler.fill_state_dict(gan.template(gan.discriminator_shapes((64, 64))))
    g, d = Gm.Generator(4, 2), Dm.Discriminator((64, 64))
    g.load_state_dict(gsd), d.load_state_dict(dsd)
    g.to(dev).train(), d.to(dev).train()
    perc = GANu.PerceptualLoss(resize_to=32, crop=28).to(dev)
    vsd = {k[len("vgg_loss.net.0."):]: v.detach().cpu().clone() for k, v in perc.state_dict().items()}
    og, od = optim.FusedAdam(g.parameters(), lr=1e-4), optim.FusedAdam(d.parameters(), lr=1e-4)
    st = recipes.GanState({k: v.clone() for k, v in gsd.items()}, {k: v.clone() for k, v in dsd.items()}, vsd, lr=1e-4,
                          vgg_resize=32, vgg_crop=28)
    lr = filler.tensor("in:gs_lr", (4, 3, 16, 16), 0.5, 0.5)
    hr = filler.tensor("in:gs_hr", (4, 3, 64, 64))
    for it in range(2):
        cap = {}
        ld, lg, fake = steps.gan_step(g, d, perc, og, od, lr.to(dev), hr.to(dev), overlap=overlap)
        rld, rlg, rfake = recipes.gan_step(st, lr, hr, capture=cap)
        assert abs(ld.item() - rld) < 0.02 * max(abs(rld), 0.1), (it, ld.item(), rld)
        assert abs(lg.item() - rlg) < 0.02 * max(abs(rlg), 0.1), (it, lg.item(), rlg)
        assert abs(losses.psnr(fake.cpu(), hr) - losses.psnr(rfake, hr)) <= 0.02
        if it == 0:       # the gradients both Adam steps consumed, from identical weights, against the bf16 floor
            from parity_util import compare_grads, prelu_ok
            sim_cap = {}
            sim = recipes.GanState({k: v.clone() for k, v in gsd.items()}, {k: v.clone() for k, v in dsd.items()}, vsd,
                                   lr=1e-4, vgg_resize=32, vgg_crop=28)
            with lowp.storage(torch.bfloat16):
                recipes.gan_step(sim, lr, hr, capture=sim_cap)
            bad_g, _ = compare_grads(dict((k, p_.grad) for k, p_ in g.named_parameters()), cap["g_grads"], sim_cap["g_grads"], "G:")
            bad_d, _ = compare_grads(dict((k, p_.grad) for k, p_ in d.named_parameters()), cap["d_grads"], sim_cap["d_grads"], "D:")
            assert not (bad_g + bad_d), bad_g + bad_d
            assert not prelu_ok(dict((k, p_.grad) for k, p_ in g.named_parameters()), cap["g_grads"], sim_cap["g_grads"])
    for mod, osd in ((g, st.g), (d, st.d)):
        for k, v in mod.state_dict().items():
            if "running_" in k:          # (measured 0.1-1.1 %: the deepest D block normalises 4x4 maps of a batch of 4 in bf16)
                assert rel_err(v.cpu(), osd[k]) < 1.5e-2, k
            if "num_batches" in k:
                assert int(v) == int(osd[k]), k
    for k, v in d.state_dict().items():          # three D forwards per step update the running statistics 3x
        if "num_batches" in k:
            assert int(v) == 6 == int(st.d[k])


@pytest.mark.parametrize("overlap", [True, False])
def test_graphed_gan_step_equals_eager(dev, overlap):
    """The whole train_GAN.py:38-71 step -- both HIP streams of steps.gan_step, three discriminator forwards, two
    backward passes, both fused Adam updates with their device-side step counters, weight re-packing, every BatchNorm
    running-statistic update -- captured ONCE in a HIP graph (steps.GraphedStep, what bench.py replays for config 3) must
    leave bit for bit the state that the same number of eager steps leaves."""
    Gm, Dm, GANu, optim, steps = P("models.GAN.generator"), P("models.GAN.discriminator"), P("utils.GAN"), P("optim"), P("steps")
    gsd = filler.fill_state_dict(gan.template(gan.generator_shapes(4, 2)))
    dsd = filler.fill_state_dict(gan.template(gan.discriminator_shapes((64, 64))))
    lr = filler.tensor("in:gg_lr", (4, 3, 16, 16), 0.5, 0.5).to(dev)
    hr = filler.tensor("in:gg_hr", (4, 3, 64, 64)).to(dev)
    perc = GANu.PerceptualLoss(resize_to=32, crop=28).to(dev)

    def make():
        g, d = Gm.Generator(4, 2), Dm.Discriminator((64, 64))
        g.load_state_dict(gsd), d.load_state_dict(dsd)
        g.to(dev).train(), d.to(dev).train()
        og, od = optim.FusedAdam(g.parameters(), lr=1e-3), optim.FusedAdam(d.parameters(), lr=1e-3)
        return g, d, (lambda: steps.gan_step(g, d, perc, og, od, lr, hr, overlap=overlap))

    g_e, d_e, step_e = make()
    for _ in range(5):
        out_e = step_e()
    g_g, d_g, step_g = make()
    graphed = steps.GraphedStep(step_g, warmup=2)         # 2 eager warm-up steps; capture itself executes nothing
    for _ in range(3):
        out_g = graphed()
    torch.cuda.synchronize()
    for a, b in zip(out_e, out_g):
        assert torch.equal(a, b)
    for me, mg in ((g_e, g_g), (d_e, d_g)):
        for (k, a), (_, b) in zip(me.state_dict().items(), mg.state_dict().items()):
            assert torch.equal(a, b), k


@pytest.mark.parametrize("graph", [False, True])
def test_fused_dense_adam_equals_separate_launches(dev, graph):
    """optim.FusedAdam(fuse_dense_head=True): dense1.weight's gradient travels as its two 16-bit factors and
    dsr_linear_wgrad_adam applies Adam inside the contraction (train_GAN.py:52-53 on discriminator.py:54 without the
    2.1 GB .grad round trip).  Same MFMA fragments in the same order and one shared statement of the Adam arithmetic:
    after 4 steps every tensor of both networks and both optimisers must equal the two-launch path bit for bit (also when
    the step replays from a HIP graph); ``.grad`` of that tensor stays None."""
    Gm, Dm, GANu, optim, steps = P("models.GAN.generator"), P("models.GAN.discriminator"), P("utils.GAN"), P("optim"), P("steps")
    gsd = filler.fill_state_dict(gan.template(gan.generator_shapes(4, 2)))
    dsd = filler.fill_state_dict(gan.template(gan.discriminator_shapes((64, 64))))
    lr = filler.tensor("in:fa_lr", (4, 3, 16, 16), 0.5, 0.5).to(dev)
    hr = filler.tensor("in:fa_hr", (4, 3, 64, 64)).to(dev)
    perc = GANu.PerceptualLoss(resize_to=32, crop=28).to(dev)

    def run(fuse):
        g, d = Gm.Generator(4, 2), Dm.Discriminator((64, 64))
        g.load_state_dict(gsd), d.load_state_dict(dsd)
        g.to(dev).train(), d.to(dev).train()
        og = optim.FusedAdam(g.parameters(), lr=1e-3)
        od = optim.FusedAdam(d.parameters(), lr=1e-3, fuse_dense_head=fuse)
        step = lambda: steps.gan_step(g, d, perc, og, od, lr, hr, overlap=True)
        if graph:
            step = steps.GraphedStep(step, warmup=2)
            n = 2
        else:
            n = 4
        for _ in range(n):
            out = step()
        torch.cuda.synchronize()
        assert (d.dense1.weight.grad is None) == fuse
        return g, d, od, out

    g_a, d_a, od_a, out_a = run(False)
    g_b, d_b, od_b, out_b = run(True)
    for a, b in zip(out_a, out_b):
        assert torch.equal(a, b)
    for ma, mb in ((g_a, g_b), (d_a, d_b)):
        for (k, a), (_, b) in zip(ma.state_dict().items(), mb.state_dict().items()):
            assert torch.equal(a, b), k
    for a, b in zip(od_a.m + od_a.v, od_b.m + od_b.v):
        assert torch.equal(a, b)
    assert int(od_a.step_t) == int(od_b.step_t) == 4
    # the bf16 shadow of dense1.weight (the next forward's MFMA operand) was refreshed by the fused launch too
    F = P("functional")
    sh = F.shadow_for_update(d_b.dense1.weight)
    assert sh is not None and torch.equal(sh, d_b.dense1.weight.detach().to(torch.bfloat16))


def test_dense_head_factors_accumulate_like_autograd(dev):
    """Two backward passes between zero_grad() and step() (gradient accumulation): the factored gradients are materialised
    and summed, exactly what autograd's .grad accumulation gives the two-launch optimiser."""
    Dm, optim, F = P("models.GAN.discriminator"), P("optim"), P("functional")
    dsd = filler.fill_state_dict(gan.template(gan.discriminator_shapes((32, 32))))
    xs = [filler.tensor(f"in:acc{i}", (2, 3, 32, 32)).to(dev) for i in range(2)]

    def run(fuse):
        d = Dm.Discriminator((32, 32))
        d.load_state_dict(dsd)
        d.to(dev).train()
        od = optim.FusedAdam(d.parameters(), lr=1e-3, fuse_dense_head=fuse)
        od.zero_grad()
        for x in xs:
            F.bce_const(d(x), 1.0).backward()
        od.step()
        torch.cuda.synchronize()
        return d

    a, b = run(False), run(True)
    for (k, u), (_, v) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.equal(u, v), k


def test_dip_step_vs_oracle(dev):
    M, D, steps = P("models.DIP"), P("utils.downsampler"), P("steps")
    kw = dict(skip_n33d=16, skip_n33u=16, skip_n11=4, num_scales=3)
    cfg = dip.SkipConfig(input_depth=8, **kw)
    sd = filler.fill_state_dict(gan.template(dip.skip_shapes(cfg)))
    net = M.get_net(8, "skip", "reflection", upsample_mode="bilinear", **kw)
    net.load_state_dict(sd)
    net.to(dev).train()
    down = D.Downsampler(3, 2, "lanczos2", phase=0.5, preserve_size=True).to(dev)
    hr = filler.tensor("in:ds_hr", (1, 3, 32, 32), 0.5, 0.5)
    lr_img = downsampler.downsampler_forward(hr, 2, "lanczos2", phase=0.5, preserve_size=True)
    zin = filler.tensor("in:ds_z", (1, 8, 32, 32), 0.05, 0.05)
    run = steps.DipRunner(net, down, zin.to(dev), lr_img.to(dev), 0.01, 0.05)
    assert run.loss_scale == 1024.0
    st = recipes.DipState({k: v.clone() for k, v in sd.items()}, cfg, zin.clone(), factor=2, lr=0.01, reg_noise_std=0.05)
    sim = recipes.DipState({k: v.clone() for k, v in sd.items()}, cfg, zin.clone(), factor=2, lr=0.01, reg_noise_std=0.05)
    for it in range(3):
        noise = filler.tensor(f"in:ds_noise{it}", (1, 8, 32, 32), 1.7)
        loss, out = run.step(noise.to(dev))
        rloss, rout = recipes.dip_step(st, lr_img, noise)
        with lowp.storage(torch.float16):
            sloss, sout = recipes.dip_step(sim, lr_img, noise)
        # floor-relative, like the outputs below: the fp16-storage oracle itself is 0.6 % / 2 % off the fp32 one after two /
        # three steps of this chaotic trajectory (lr 0.01), and a last-bit change of the target image moves the HIP loss of the
        # third step by 2 %
        assert abs(loss.item() - rloss) < max(0.03 * abs(rloss), 3.0 * abs(float(sloss) - float(rloss))), (it, loss.item(), rloss, float(sloss))
        floor = (sout - rout).abs().max().item()
        assert (out.cpu() - rout).abs().max().item() <= 3.0 * floor + 0.01, (it, floor)


def test_dip_graphed_iteration_equals_eager(dev):
    """A Deep-Image-Prior iteration (DIP.py:47-68 closure + Adam) replayed from a HIP graph leaves the same state as eager
    iterations (reg_noise_std = 0 so that no random numbers are drawn; with noise the graph draws fresh noise per replay
    from the registered generator, which has no eager twin to compare with bit for bit)."""
    M, D, steps = P("models.DIP"), P("utils.downsampler"), P("steps")
    kw = dict(skip_n33d=16, skip_n33u=16, skip_n11=4, num_scales=3)
    sd = filler.fill_state_dict(gan.template(dip.skip_shapes(dip.SkipConfig(input_depth=8, **kw))))
    down = D.Downsampler(3, 2, "lanczos2", phase=0.5, preserve_size=True).to(dev)
    hr = filler.tensor("in:dg_hr", (1, 3, 32, 32), 0.5, 0.5)
    lr_img = downsampler.downsampler_forward(hr, 2, "lanczos2", phase=0.5, preserve_size=True).to(dev)
    zin = filler.tensor("in:dg_z", (1, 8, 32, 32), 0.05, 0.05)

    def make():
        net = M.get_net(8, "skip", "reflection", upsample_mode="bilinear", **kw)
        net.load_state_dict(sd)
        net.to(dev).train()
        return net, steps.DipRunner(net, down, zin.to(dev), lr_img, 0.01, 0.0)

    net_e, run_e = make()
    for _ in range(4):
        loss_e, out_e = run_e.step()
    net_g, run_g = make()
    graphed = steps.GraphedStep(run_g.step, warmup=2)
    for _ in range(2):
        loss_g, out_g = graphed()
    torch.cuda.synchronize()
    assert torch.equal(loss_e, loss_g) and torch.equal(out_e, out_g)
    for (k, a), (_, b) in zip(net_e.state_dict().items(), net_g.state_dict().items()):
        assert torch.equal(a, b), k


def test_two_rank_gan_step_rehearsal(dev):
    """The N > 1 path end to end on the GPU: two ranks share cuda:0 and talk over gloo (RCCL cannot place two ranks
    on one device), running bench.py's config-3 step -- parameter broadcast, gradient hooks for the 2 GB dense1
    gradient (all-gathered rank-local factors + dsr_linear_wgrad_gathered, and the plain all-reduce it replaces),
    bucketed all-reduce, the two-stream D/G overlap.  bench.py itself asserts that both ranks hold bit-identical
    parameters after the averaged updates; here the two exchange forms must also agree with each other."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DSR_DIST_REHEARSAL="1", MASTER_ADDR="127.0.0.1")
    import socket
    with socket.socket() as sk:                      # any free port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
           "--no-cpu-baseline", "--no-roofline", "--no-other-workloads"]
    import json
    import re
    sums = {}
    for mode in ("1", "0"):      # dense1 gradient by all-gathered factors (default) | by plain all-reduce
        r = subprocess.run(cmd, cwd=root, env=dict(env, DSR_DP_FACTOR_GATHER=mode), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-6000:]
        assert "rehearsal: parameters identical on all ranks" in r.stderr
        out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        assert out["n_gpus"] == 2 and out["value"] > 0
        sums[mode] = {m.group(1): float(m.group(2)) for m in re.finditer(r"checksum (\w+) ([0-9.e+-]+)", r.stderr)}
    # both exchanges produce the same averaged update (they differ only in fp32 summation order)
    for name, v in sums["1"].items():
        assert abs(v - sums["0"][name]) <= 1e-6 * abs(v), (name, v, sums["0"][name])


@pytest.mark.parametrize("fuse", [True, False], ids=["fused_dense_adam", "materialised_dense_grad"])
def test_two_rank_step_vs_k_shard_oracle(dev, fuse, tmp_path):
    """the parity oracle of data parallelism is a single-process k-shard emulation.  Two ranks (one GPU, gloo)
    run ONE gan_step each on their half of a 4-sample batch from the same closed-form weights (tests/dp_shard_worker.py: the
    hooks, buckets, dense-head factor gather and fused dense-head Adam of the N > 1 path).  The oracle runs
    recipes.gan_step's forward / backward on each shard from those weights, the captured gradients are AVERAGED over the
    shards and one Adam step (torch.optim.Adam defaults, t = 1) is applied on the CPU.  Compared: every averaged gradient
    tensor by the floor rule of tests/parity_util.py (floor = the same emulation with bf16 conv storage); every parameter's
    displacement against the oracle's (cosine >= 0.9 for tensors of >= 4096 elements whose oracle gradient is not vanishing,
    and never longer than lr per element); both ranks end with bit-identical parameters; BatchNorm running statistics stay
    RANK-LOCAL (each rank's equal its own shard's oracle statistics: dist.py's stated semantics, DDP's convention)."""
    import os
    import socket
    import subprocess
    import sys
    from parity_util import compare_grads, cos as pcos
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "tests", "dp_shard_worker.py"), str(tmp_path), "1" if fuse else "0"]
    r = subprocess.run(cmd, cwd=root, env=dict(os.environ, MASTER_ADDR="127.0.0.1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-6000:]
    ranks = [torch.load(os.path.join(str(tmp_path), f"rank{i}.pt"), weights_only=True) for i in range(2)]
    # ---- both ranks applied the same update
    for net in ("g", "d"):
        for k, v in ranks[0][net].items():
            if "running_" in k or "num_batches" in k:
                continue
            assert torch.equal(v, ranks[1][net][k]), (net, k)
    for k, v in ranks[0]["grads"].items():
        assert torch.equal(v, ranks[1]["grads"][k]), k
    # ---- k-shard emulation on the CPU
    GANu = P("utils.GAN")
    gsd = filler.fill_state_dict(gan.template(gan.generator_shapes(4, 2)))
    dsd = filler.fill_state_dict(gan.template(gan.discriminator_shapes((64, 64))))
    perc = GANu.PerceptualLoss(resize_to=32, crop=28)
    vsd = {k[len("vgg_loss.net.0."):]: v.detach().clone() for k, v in perc.state_dict().items()}
    lr = filler.tensor("in:dp_lr", (4, 3, 16, 16), 0.5, 0.5)
    hr = filler.tensor("in:dp_hr", (4, 3, 64, 64))

    def emulate(storage):
        caps, states = [], []
        for s_ in range(2):
            st = recipes.GanState({k: v.clone() for k, v in gsd.items()}, {k: v.clone() for k, v in dsd.items()}, vsd, lr=1e-4,
                                  vgg_resize=32, vgg_crop=28)
            cap = {}
            if storage is None:
                out = recipes.gan_step(st, lr[2 * s_:2 * s_ + 2], hr[2 * s_:2 * s_ + 2], capture=cap)
            else:
                with lowp.storage(storage):
                    out = recipes.gan_step(st, lr[2 * s_:2 * s_ + 2], hr[2 * s_:2 * s_ + 2], capture=cap)
            caps.append(cap)
            states.append((st, out, cap.get("content")))
        avg = {}
        for tag, key in (("G:", "g_grads"), ("D:", "d_grads")):
            for k in caps[0][key]:
                if caps[0][key][k] is not None:
                    avg[tag + k] = (caps[0][key][k] + caps[1][key][k]) / 2
        return avg, states
    ref, ref_states = emulate(None)
    sim, _ = emulate(torch.bfloat16)
    hip = ranks[0]["grads"]
    assert ("D:dense1.weight" in hip) == (not fuse)          # the fused form never materialises that gradient
    strip = lambda d_, t: {k[2:]: v for k, v in d_.items() if k.startswith(t)}
    bad = []
    for t in ("G:", "D:"):
        b, _ = compare_grads(strip(hip, t), strip(ref, t), strip(sim, t), t)
        bad += b
    assert not bad, bad
    # ---- one Adam step from zero moments on the averaged gradient: p - lr * g / (|g| + eps)  (m_hat = g, v_hat = g^2)
    # (Adam's first step is -lr * sign(g) wherever |g| >> eps: the displacement cosine counts sign agreements, so it is judged
    #  against the same number of the bf16-storage emulation, like the gradients themselves)
    for net, init, tag in (("g", gsd, "G:"), ("d", dsd, "D:")):
        for k, p0 in init.items():
            gk = ref.get(tag + k)
            if gk is None or not p0.dtype.is_floating_point or "running_" in k:
                continue
            step_ref = -1e-4 * gk / (gk.abs() + 1e-8)
            step_hip = ranks[0][net][k] - p0
            assert float(step_hip.abs().max()) <= 1.0001e-4 + 1e-7 * float(p0.abs().max()), (net, k)
            if gk.numel() >= 4096 and not pre_bn_bias_(k):
                gs = sim[tag + k]
                c, cf = pcos(step_hip, step_ref), pcos(-1e-4 * gs / (gs.abs() + 1e-8), step_ref)
                assert (1 - c) <= 1.5 * (1 - cf) + 0.02 and c >= 0.8, (net, k, c, cf)
    # ---- BatchNorm statistics are per rank: rank r's equal the oracle's run on shard r
    for r_ in range(2):
        st = ref_states[r_][0]
        for net, osd in (("g", st.g), ("d", st.d)):
            for k, v in ranks[r_][net].items():
                if "running_" in k:
                    assert rel_err(v, osd[k]) < 2e-2, (r_, net, k)
    assert not torch.equal(ranks[0]["g"]["bn1.running_mean"], ranks[1]["g"]["bn1.running_mean"])
    # ---- and the losses each rank reports are its shard's: loss_D directly; loss_G = the shard's content loss + the adversarial
    # number of the discriminator AFTER the rank-averaged Adam step (train_GAN.py:53,58-59), on the shard's generated images.
    # One Adam step moves every discriminator weight by +-lr on the sign of its gradient, so that number is only as close as the
    # sign patterns are (the displacement check above): 10 % of the loss
    d_after = {k: v.clone() for k, v in dsd.items()}
    for k in d_after:
        gk = ref.get("D:" + k)
        if gk is not None and d_after[k].dtype.is_floating_point and "running_" not in k:
            d_after[k] = d_after[k] - 1e-4 * gk / (gk.abs() + 1e-8)
    for r_ in range(2):
        _, (rld, _, rfake), content = ref_states[r_]
        assert abs(ranks[r_]["loss_d"] - rld) <= 0.02 * max(abs(rld), 0.1)
        with torch.no_grad():
            adv = float(losses.adversarial(gan.discriminator_forward({k: v.clone() for k, v in d_after.items()}, rfake, True)))
        rlg = content + adv
        assert abs(ranks[r_]["loss_g"] - rlg) <= 0.10 * max(abs(rlg), 0.1), (ranks[r_]["loss_g"], rlg, content, adv)


def pre_bn_bias_(k):
    from parity_util import pre_bn_bias
    return pre_bn_bias(k)


def test_rccl_single_rank_gan_step(dev):
    """The same N > 1 code path through RCCL itself (backend "nccl"), which a one-GPU box can only run as a world of ONE
    rank (DSR_DIST_FORCE=1): broadcast, bucketed all-reduce (ReduceOp.AVG) into the persistent bucket views, the factor
    all-gather / the hook-issued 2 GB all-reduce, all on the two streams of the overlapped step.  With one rank the
    averaged update IS the local update, so the parameters must match a run without any process group."""
    import json
    import os
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-roofline",
           "--no-other-workloads"]
    sums = {}
    # (plain: eager like the two distributed runs -- a graph-replayed bench takes three more warm-up steps)
    for tag, extra in (("plain", dict(DSR_BENCH_CHECKSUM="1", DSR_GAN_GRAPH="0")), ("gather", dict(DSR_DIST_FORCE="1")),
                       ("allreduce", dict(DSR_DIST_FORCE="1", DSR_DP_FACTOR_GATHER="0"))):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", **extra)
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
            env.pop(k, None)
        r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-6000:]
        out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        assert out["n_gpus"] == 1 and out["value"] > 0
        sums[tag] = {m.group(1): float(m.group(2)) for m in re.finditer(r"checksum (\w+) ([0-9.e+-]+)", r.stderr)}
        assert set(sums[tag]) == {"Generator", "Discriminator"}, r.stderr[-2000:]
    for tag in ("gather", "allreduce"):
        for name, v in sums["plain"].items():
            assert abs(v - sums[tag][name]) <= 1e-6 * abs(v), (tag, name, v, sums[tag][name])


def test_full_size_batch_split_invariance(dev):
    """BASELINE config-3 sizes (batch 32, 128x128 -> 512x512), where no CPU oracle finishes in test time: in eval mode
    every image is independent, so the networks run on the whole batch (the launch shapes bench.py times: persistent
    and sliced kernels, 16k-tile grids, the 524,288-feature dense head) must reproduce, image for image, what they
    produce on one image at a time (small grids, one-tile-per-block kernels).  Bit for bit for the generator and
    the VGG trunk; the discriminator's dense head sums its split-K slabs in a batch-dependent order (1e-5)."""
    Gm, Dm, GANu = P("models.GAN.generator"), P("models.GAN.discriminator"), P("utils.GAN")
    torch.manual_seed(3)
    lr = torch.rand(32, 3, 128, 128, device=dev)
    with torch.no_grad():
        gen = Gm.Generator(4, 16).to(dev).eval()
        sr = gen(lr)
        assert tuple(sr.shape) == (32, 3, 512, 512) and torch.isfinite(sr).all()
        for i in (0, 13, 31):
            assert torch.equal(sr[i:i + 1], gen(lr[i:i + 1].contiguous())), i
        disc = Dm.Discriminator((512, 512)).to(dev).eval()
        p = disc(sr)
        assert tuple(p.shape) == (32, 1) and torch.isfinite(p).all()
        for i in (0, 31):
            assert (p[i:i + 1] - disc(sr[i:i + 1].contiguous())).abs().max().item() <= 1e-5, i
        vgg = GANu.Vgg19Loss().to(dev)
        f = vgg.features(sr)
        assert f.shape[0] == 32 and torch.isfinite(f).all()
        for i in (0, 17):
            assert torch.equal(f[i:i + 1], vgg.features(sr[i:i + 1].contiguous())), i


def test_full_size_gan_step_bookkeeping(dev):
    """One config-3 step at full size (batch 32, 128x128 -> 512x512, two-stream form): finite losses, every weight tensor of
    both networks moved and everything stayed finite, BatchNorm counters advanced as train_GAN.py:44-58 implies (generator twice,
    discriminator three times), and the single-stream form of the same step from the same state gives the same
    numbers (the streams only reorder independent work)."""
    Gm, Dm, GANu, steps, optim = (P("models.GAN.generator"), P("models.GAN.discriminator"), P("utils.GAN"), P("steps"),
                                  P("optim"))
    torch.manual_seed(5)
    lr = torch.rand(32, 3, 128, 128, device=dev)
    hr = torch.rand(32, 3, 512, 512, device=dev) * 2 - 1
    perc = GANu.PerceptualLoss().to(dev)
    results = []
    for overlap in (True, False):
        torch.manual_seed(11)
        gen = Gm.Generator(4, 16).to(dev).train()
        disc = Dm.Discriminator((512, 512)).to(dev).train()
        # (conv biases in front of a train-mode BatchNorm have an exactly zero gradient and stay put)
        g0 = {k: p.detach().clone() for k, p in gen.named_parameters() if k.endswith("weight")}
        d0 = {k: p.detach().clone() for k, p in disc.named_parameters() if k.endswith("weight") and "dense1" not in k}
        og, od = optim.FusedAdam(gen.parameters(), lr=1e-4), optim.FusedAdam(disc.parameters(), lr=1e-4)
        ld, lg, fake = steps.gan_step(gen, disc, perc, og, od, lr, hr, overlap=overlap)
        torch.cuda.synchronize()
        assert torch.isfinite(ld) and torch.isfinite(lg) and 0.0 < ld.item() < 20.0
        assert all(torch.isfinite(p).all() for p in gen.parameters()) and all(torch.isfinite(p).all() for p in disc.parameters())
        assert all(not torch.equal(g0[k], p) for k, p in gen.named_parameters() if k in g0)
        assert all(not torch.equal(d0[k], p) for k, p in disc.named_parameters() if k in d0)
        assert all(int(v) == 2 for k, v in gen.state_dict().items() if k.endswith("num_batches_tracked"))
        assert all(int(v) == 3 for k, v in disc.state_dict().items() if k.endswith("num_batches_tracked"))
        results.append((ld.item(), lg.item(), float(disc.dense1.weight.detach().double().sum()), float(fake.double().sum())))
        del gen, disc, og, od
    a, b = results
    assert all(abs(x - y) <= 1e-6 * max(1.0, abs(x)) for x, y in zip(a, b)), (a, b)


@pytest.mark.parametrize("overlap", [True, False])
def test_batched_wgrad_step_equals_per_layer_step(dev, overlap):
    """steps.gan_step with functional.batched_wgrad (one grouped weight-gradient launch per backward pass; the
    discriminator's real + generated contributions summed inside its reduction) against the same step with one launch per
    layer and autograd's own accumulation: the same gradients up to fp32 summation order, so after one step every parameter
    moves by the same Adam update to ~1e-6 (a flipped sign of a near-zero gradient would show as 2 lr = 2e-3)."""
    Gm, Dm, GANu, optim, steps = P("models.GAN.generator"), P("models.GAN.discriminator"), P("utils.GAN"), P("optim"), P("steps")
    gsd = filler.fill_state_dict(gan.template(gan.generator_shapes(4, 2)))
    dsd = filler.fill_state_dict(gan.template(gan.discriminator_shapes((64, 64))))
    lr = filler.tensor("in:bw_lr", (4, 3, 16, 16), 0.5, 0.5).to(dev)
    hr = filler.tensor("in:bw_hr", (4, 3, 64, 64)).to(dev)
    perc = GANu.PerceptualLoss(resize_to=32, crop=28).to(dev)
    F = P("functional")

    def run(batch):
        g, d = Gm.Generator(4, 2), Dm.Discriminator((64, 64))
        g.load_state_dict(gsd), d.load_state_dict(dsd)
        g.to(dev).train(), d.to(dev).train()
        og, od = optim.FusedAdam(g.parameters(), lr=1e-3), optim.FusedAdam(d.parameters(), lr=1e-3)
        out = steps.gan_step(g, d, perc, og, od, lr, hr, overlap=overlap, batch_wgrad=batch)
        torch.cuda.synchronize()
        grads = {("g." + k): p.grad.detach().clone() for k, p in g.named_parameters() if p.grad is not None}
        grads.update({("d." + k): p.grad.detach().clone() for k, p in d.named_parameters() if p.grad is not None})
        return g, d, out, grads

    g_a, d_a, out_a, gr_a = run(False)
    g_b, d_b, out_b, gr_b = run(True)
    assert F._wgrad_batch is None
    assert set(gr_a) == set(gr_b)
    for k in gr_a:
        a, b = gr_a[k].double(), gr_b[k].double()
        assert torch.isfinite(b).all(), k
        assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max()) + 1e-12, k
    # loss_D and the generator output are formed before any update: equal to fp32 summation order.  loss_G contains the
    # adversarial number of the discriminator AFTER its Adam step (lr 1e-3 here): a gradient element within 1e-5 of zero may
    # move its weight by +lr in one run and -lr in the other, which shows in that number at the 1e-4 level
    assert torch.allclose(out_a[0], out_b[0], rtol=1e-5, atol=1e-7) and torch.allclose(out_a[2], out_b[2], rtol=1e-5, atol=1e-7)
    assert torch.allclose(out_a[1], out_b[1], rtol=5e-4, atol=1e-6)
    for ma, mb in ((g_a, g_b), (d_a, d_b)):
        for (k, a), (_, b) in zip(ma.state_dict().items(), mb.state_dict().items()):
            if a.dtype.is_floating_point and not k.endswith(("running_mean", "running_var")):
                # Adam's first step moves a parameter by lr * g / (|g| + eps): only where |g| ~ eps = 1e-8 can a 1e-6
                # relative change of g show; allow a few such elements
                bad = ((a - b).abs() > 2e-5).float().mean().item()
                assert bad < 1e-3, (k, bad)


def test_batched_wgrad_leaves_existing_grad_and_exceptions_alone(dev):
    """A weight whose .grad is already set (accumulation over two backward passes) is not batched -- autograd adds into it at
    once, which needs the finished gradient -- and an exception inside the block discards the batch."""
    Gm, F = P("models.GAN.generator"), P("functional")
    gsd = filler.fill_state_dict(gan.template(gan.generator_shapes(4, 2)))
    x = filler.tensor("in:bw2", (2, 3, 16, 16), 0.5, 0.5).to(dev)
    hr = filler.tensor("in:bw2_hr", (2, 3, 64, 64)).to(dev)

    def grads(two_pass, batch):
        g = Gm.Generator(4, 2)
        g.load_state_dict(gsd)
        g.to(dev).train()
        for _ in range(2 if two_pass else 1):
            with F.batched_wgrad(batch):
                F.l1_loss(g(x), hr).backward()
        torch.cuda.synchronize()
        return {k: p.grad.detach().clone() for k, p in g.named_parameters() if p.grad is not None}

    a, b = grads(True, False), grads(True, True)
    for k in a:
        assert float((a[k].double() - b[k].double()).abs().max()) <= 2e-5 * float(a[k].abs().max()) + 1e-12, k
    with pytest.raises(ZeroDivisionError):
        with F.batched_wgrad():
            1 / 0
    assert F._wgrad_batch is None


def test_batched_wgrad_early_launches_on_another_stream(dev):
    """functional.batched_wgrad(early_stream=, early_every=): every few collected layers are launched at once on another stream
    (behind an event of the backward pass's stream) instead of at the exit -- steps.gan_step does that for the generator half.
    The same gradients as the exit launch up to fp32 summation order (another split of the pixels over blocks), for a group
    size that divides the layer count, one that does not, and one larger than it (nothing early); a weight used a second time
    after its early launch (the same generator applied twice in one graph) is still summed correctly."""
    Gm, F = P("models.GAN.generator"), P("functional")
    gsd = filler.fill_state_dict(gan.template(gan.generator_shapes(4, 3)))
    x = filler.tensor("in:bw_early", (2, 3, 16, 16), 0.5, 0.5).to(dev)
    hr = filler.tensor("in:bw_early_hr", (2, 3, 64, 64)).to(dev)
    side = torch.cuda.Stream()

    def grads(every, twice):
        g = Gm.Generator(4, 3)
        g.load_state_dict(gsd)
        g.to(dev).train()
        with F.batched_wgrad(True, early_stream=side if every else None, early_every=every) as ctx:
            loss = F.l1_loss(g(x), hr)
            if twice:
                loss = F.add_losses(loss, F.l1_loss(g(x * 0.5), hr))
            loss.backward()
            launched_early = len(ctx.done)
        torch.cuda.synchronize()
        return {k: p.grad.detach().clone() for k, p in g.named_parameters() if p.grad is not None}, launched_early

    for twice in (False, True):
        ref, n0 = grads(0, twice)
        assert n0 == 0
        for every in (2, 3, 100):
            got, n_early = grads(every, twice)
            assert (n_early > 0) == (every < 100), (every, n_early)
            assert set(got) == set(ref)
            for k in ref:
                a, b = ref[k].double(), got[k].double()
                assert torch.isfinite(b).all() and float((a - b).abs().max()) <= 2e-5 * float(a.abs().max()) + 1e-12, (twice, every, k)
    assert F._wgrad_batch is None


def test_batched_wgrad_weight_with_a_batchable_and_an_unbatchable_use(dev):
    """One 3x3 weight applied twice in a graph, once at stride 1 (the grouped launch takes it) and once at stride 2 (it does
    not), in both orders of the backward pass: inside functional.batched_wgrad the first use's placeholder must not be summed
    with the second use's real tensor.  Weight, bias and input gradients equal the unbatched run's."""
    F = P("functional")
    w0 = bfr(filler.tensor("w:mixed", (64, 64, 3, 3), 0.05))
    x0 = to_nhwc(bfr(filler.tensor("x:mixed", (2, 64, 16, 24)))).to(dev)
    res = {}
    for order in ("s1_first", "s2_first"):
        for batch in (False, True):
            w = w0.clone().to(dev).requires_grad_(True)
            x = x0.clone().requires_grad_(True)
            strides = (1, 2) if order == "s1_first" else (2, 1)
            ya = F.ConvAct.apply(x, w, None, None, dict(stride=strides[0], pad=1))
            yb = F.ConvAct.apply(x, w, None, None, dict(stride=strides[1], pad=1))
            loss = ya.float().square().mean() + yb.float().square().mean()
            with F.batched_wgrad(batch) as ctx:
                loss.backward()
                if batch:
                    assert len(ctx.items) + len(ctx.unbatched) == 1
            torch.cuda.synchronize()
            assert torch.isfinite(w.grad).all()
            res[(order, batch)] = (w.grad.clone(), x.grad.clone())
        a, b = res[(order, False)], res[(order, True)]
        assert float((a[0] - b[0]).abs().max()) <= 1e-5 * float(a[0].abs().max()), order
        assert torch.equal(a[1], b[1])
    assert F._wgrad_batch is None


def test_batched_wgrad_exception_drops_placeholder_grads(dev):
    """If backward raises inside the block the grouped launch never runs: a .grad that already points at a placeholder is
    removed instead of being left as uninitialised memory."""
    F = P("functional")
    w = bfr(filler.tensor("w:exc", (64, 64, 3, 3), 0.05)).to(dev).requires_grad_(True)
    x = to_nhwc(bfr(filler.tensor("x:exc", (1, 64, 8, 8)))).to(dev)
    y = F.ConvAct.apply(x, w, None, None, dict(stride=1, pad=1))
    with pytest.raises(ZeroDivisionError):
        with F.batched_wgrad():
            y.float().sum().backward()
            assert w.grad is not None
            1 / 0
    assert w.grad is None and F._wgrad_batch is None


def test_vgg_trunk_act_links_equal_separate_passes(dev):
    """utils.GAN.Vgg19Loss with functional.ActLink (every ReLU's backward folded into its consumer's input-gradient / max-pool
    backward launch) against the same trunk with one activation-backward pass per layer: loss and image gradient bit for bit."""
    GANu, F = P("utils.GAN"), P("functional")
    loss_mod = GANu.Vgg19Loss().to(dev)
    img = filler.tensor("in:vl_a", (2, 3, 96, 96)).to(dev)
    tgt = filler.tensor("in:vl_b", (2, 3, 96, 96)).to(dev)
    res = []
    try:
        for links in (False, True):
            F.ACT_LINKS = links
            a = img.clone().requires_grad_(True)
            loss = loss_mod(a, tgt)
            loss.backward()
            torch.cuda.synchronize()
            res.append((loss.detach().clone(), a.grad.clone()))
    finally:
        F.ACT_LINKS = True
    assert torch.isfinite(res[1][1]).all() and float(res[1][1].abs().max()) > 0
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
