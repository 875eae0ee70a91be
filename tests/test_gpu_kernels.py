"""GPU parity tests of the HIP kernels (through the C ABI) against a plain PyTorch fp32 CPU
reference of the same op on the same bf16-rounded operands.

Tolerances: operands are exactly representable in bf16 on both sides, accumulation is fp32 on
both sides, so a single op differs only by the final bf16 rounding of the output (2^-9 relative)
and by summation order.  Gradients w.r.t. weights are fp32 outputs.
"""
import importlib

import numpy as np
import pytest
import torch
import torch.nn.functional as TF

from oracle import filler

pytestmark = pytest.mark.gpu

PKG = "deep-super-resolution_amd"


def P(sub):
    return importlib.import_module(PKG + "." + sub)


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    P("_lib").lib()          # raises if the extension is missing: no silent fallback
    return torch.device("cuda:0")


def bfr(t):
    return t.to(torch.bfloat16).float()


def rel_err(got, ref):
    got, ref = got.double(), ref.double()
    return float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-20))


def to_nhwc(x, cp=None):
    n, c, h, w = x.shape
    cp = cp or (c + 7) // 8 * 8
    out = torch.zeros(n, h, w, cp, dtype=torch.bfloat16)
    out[..., :c] = x.permute(0, 2, 3, 1).to(torch.bfloat16)
    return out


def from_nhwc(y, c):
    return y[..., :c].float().permute(0, 3, 1, 2).contiguous()


def pad_ref(x, pad, mode):
    if pad == 0:
        return x
    if mode == 0:
        return TF.pad(x, (pad,) * 4)
    return TF.pad(x, (pad,) * 4, mode="reflect" if mode == 1 else "replicate")


def act_ref(F, y, act, slope):
    if act == F.ACT_NONE:
        return y
    if act in (F.ACT_LEAKY, F.ACT_PRELU):
        return torch.where(y >= 0, y, y * slope)
    if act == F.ACT_RELU:
        return torch.relu(y)
    if act == F.ACT_TANH:
        return torch.tanh(y)
    return torch.sigmoid(y)


CONV_CASES = [
    # name, N, Cin, Cout, H, W, k, stride, pad, pad_mode, act
    ("trunk3x3", 2, 64, 64, 12, 20, 3, 1, 1, 0, "none"),
    ("trunk3x3_tail", 1, 64, 64, 7, 9, 3, 1, 1, 0, "leaky"),
    ("head9x9", 2, 3, 64, 16, 16, 9, 1, 4, 0, "prelu"),
    ("head9x9_ragged", 3, 3, 64, 37, 45, 9, 1, 4, 0, "prelu"),          # conv_rgb9_kernel: 5 x 2 tiles per image, ragged right and bottom
    ("head9x9_gray", 1, 1, 64, 9, 33, 9, 1, 4, 0, "leaky"),              # ... one real input channel, a map smaller than the window's reach
    ("d_first", 2, 3, 64, 16, 24, 3, 1, 1, 0, "leaky"),
    ("d_first_ragged", 3, 3, 64, 37, 70, 3, 1, 1, 0, "leaky"),
    ("c64_wide_192", 2, 64, 192, 19, 45, 3, 1, 1, 0, "relu"),          # weights-in-registers kernel, 3 output slices
    ("ps_64_256_ragged", 2, 64, 256, 21, 75, 3, 1, 1, 0, "leaky"),       # input gradient: halo-staged 64-output kernel, 256 channels in, ragged tiles
    ("d_b1_64_128", 1, 64, 128, 16, 64, 3, 1, 1, 0, "none"),            # ... 128 channels in, exactly one tile row of 2 x 1 tiles
    ("d_s2_64", 2, 64, 64, 16, 16, 3, 2, 1, 0, "none"),
    ("d_s2_odd", 1, 64, 128, 15, 17, 3, 2, 1, 0, "none"),
    ("d_128_256", 1, 128, 256, 8, 8, 3, 1, 1, 0, "relu"),
    ("d_512", 1, 256, 512, 6, 6, 3, 1, 1, 0, "none"),
    ("dip_reflect_s2", 1, 32, 128, 16, 16, 3, 2, 1, 1, "none"),
    ("dip_reflect_132", 1, 132, 128, 10, 12, 3, 1, 1, 1, "none"),
    ("dip_reflect_128", 2, 128, 128, 13, 19, 3, 1, 1, 1, "leaky"),      # reflect padding on the LDS-DMA path (padded coordinates per K-step)
    ("dip_reflect_128_s2", 1, 128, 128, 16, 18, 3, 2, 1, 1, "none"),
    ("dip_reflect_256_192", 1, 256, 192, 9, 11, 3, 1, 1, 1, "none"),
    ("dip_1x1_skip", 1, 32, 4, 12, 12, 1, 1, 0, 1, "none"),
    ("dip_1x1_128", 2, 128, 128, 8, 8, 1, 1, 0, 0, "leaky"),
    ("cout3_9x9", 1, 64, 3, 12, 12, 9, 1, 4, 0, "none"),
    ("cout3_9x9_ragged", 2, 64, 3, 75, 140, 9, 1, 4, 0, "none"),    # Toeplitz fwd/wgrad: 3 strips (last ragged), 3 row bands
    ("vgg_3_64", 1, 3, 64, 14, 14, 3, 1, 1, 0, "relu"),
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_act_fwd_bwd(dev, case):
    F = P("functional")
    name, n, cin, cout, h, w, k, stride, pad, pmode, actn = case
    act = dict(none=F.ACT_NONE, leaky=F.ACT_LEAKY, prelu=F.ACT_PRELU, relu=F.ACT_RELU)[actn]
    slope = 0.2 if actn == "leaky" else 0.25
    x = bfr(filler.tensor("x:" + name, (n, cin, h, w)))
    wt = bfr(filler.tensor("w:" + name, (cout, cin, k, k), float(np.sqrt(3.0 / (cin * k * k)))))
    b = filler.tensor("b:" + name, (cout,), 0.1)
    # ---- CPU reference (fp32)
    xr = x.clone().requires_grad_(True)
    wr = wt.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    ar = torch.tensor([slope], requires_grad=True)
    yr = TF.conv2d(pad_ref(xr, pad, pmode), wr, br, stride=stride)
    yr = act_ref(F, yr, act, ar if act == F.ACT_PRELU else slope)
    probe = bfr(filler.tensor("p:" + name, tuple(yr.shape)))
    (yr * probe).sum().backward()
    # ---- HIP
    xg = to_nhwc(x).to(dev).requires_grad_(True)
    wg = wt.to(dev).requires_grad_(True)
    bg = b.to(dev).requires_grad_(True)
    ag = torch.tensor([slope], device=dev, requires_grad=True) if act == F.ACT_PRELU else None
    cfg = dict(stride=stride, pad=pad, pad_mode=pmode, act=act, slope=slope)
    yg = F.ConvAct.apply(xg, wg, bg, ag, cfg)
    yg.backward(to_nhwc(probe, yg.shape[-1]).to(dev))
    torch.cuda.synchronize()
    y = from_nhwc(yg.detach().cpu(), cout)
    assert rel_err(y, yr.detach()) < 1.2e-2, ("fwd", rel_err(y, yr.detach()))
    # pad channels of the output must be exactly zero (they feed the next layer's K dimension)
    if yg.shape[-1] > cout:
        assert float(yg.detach()[..., cout:].float().abs().max()) == 0.0
    # the CPU reference back-propagates the fp32 pre-rounding output; the HIP path rounds dy to bf16
    assert rel_err(from_nhwc(xg.grad.cpu(), cin), xr.grad) < 2.5e-2, ("dgrad", rel_err(from_nhwc(xg.grad.cpu(), cin), xr.grad))
    assert rel_err(wg.grad.cpu(), wr.grad) < 2.5e-2, ("wgrad", rel_err(wg.grad.cpu(), wr.grad))
    assert rel_err(bg.grad.cpu(), br.grad) < 2.5e-2, ("bias grad", rel_err(bg.grad.cpu(), br.grad))
    if ag is not None:
        assert rel_err(ag.grad.cpu(), ar.grad) < 3e-2, ("prelu grad", ag.grad.cpu(), ar.grad)


def test_conv_exact_integers(dev):
    """Small-integer operands make every product and partial sum exact: the MFMA lane maps, the LDS
    swizzle, the tap table and the epilogue must then reproduce the reference bit for bit."""
    F = P("functional")
    g = torch.Generator().manual_seed(5)
    x = torch.randint(-3, 4, (2, 64, 10, 13), generator=g).float()
    wt = torch.randint(-2, 3, (64, 64, 3, 3), generator=g).float()
    b = torch.randint(-4, 5, (64,), generator=g).float()
    yr = TF.conv2d(x, wt, b, padding=1)
    yr_b = bfr(yr)
    yg = F.ConvAct.apply(to_nhwc(x).to(dev), wt.to(dev), b.to(dev), None, dict(stride=1, pad=1, act=F.ACT_NONE))
    torch.cuda.synchronize()
    assert torch.equal(from_nhwc(yg.cpu(), 64), yr_b)


def test_pixel_shuffle_conv(dev):
    F = P("functional")
    n, h, w = 2, 6, 7
    x = bfr(filler.tensor("x:ps", (n, 64, h, w)))
    wt = bfr(filler.tensor("w:ps", (256, 64, 3, 3), float(np.sqrt(3.0 / 576))))
    b = filler.tensor("b:ps", (256,), 0.1)
    xr, wr, br = x.clone().requires_grad_(True), wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ar = torch.tensor([0.25], requires_grad=True)
    yr = TF.pixel_shuffle(TF.conv2d(xr, wr, br, padding=1), 2)
    yr = torch.where(yr >= 0, yr, ar * yr)
    probe = bfr(filler.tensor("p:ps", tuple(yr.shape)))
    (yr * probe).sum().backward()
    xg = to_nhwc(x).to(dev).requires_grad_(True)
    wg, bg = wt.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    ag = torch.tensor([0.25], device=dev, requires_grad=True)
    yg = F.ConvAct.apply(xg, wg, bg, ag, dict(stride=1, pad=1, act=F.ACT_PRELU, pixel_shuffle=True))
    assert tuple(yg.shape) == (n, 2 * h, 2 * w, 64)
    yg.backward(to_nhwc(probe).to(dev))
    torch.cuda.synchronize()
    assert rel_err(from_nhwc(yg.detach().cpu(), 64), yr.detach()) < 1.2e-2
    assert rel_err(from_nhwc(xg.grad.cpu(), 64), xr.grad) < 2.5e-2
    assert rel_err(wg.grad.cpu(), wr.grad) < 2.5e-2
    assert rel_err(bg.grad.cpu(), br.grad) < 2.5e-2
    assert rel_err(ag.grad.cpu(), ar.grad) < 3e-2


BN_CASES = [
    ("g_bn_prelu", 2, 64, 64, 10, 12, 1, 0, "prelu", False, True),
    ("g_bn_res", 2, 64, 64, 10, 12, 1, 0, "none", True, True),
    ("d_bn_leaky_s2", 3, 64, 128, 12, 12, 2, 0, "leaky", False, True),
    ("d_bn_64_128_s1", 2, 64, 128, 21, 37, 1, 0, "leaky", False, True),   # sliced c64 kernel with statistics rows
    ("dip_bn_reflect", 1, 32, 128, 16, 16, 2, 1, "leaky", False, True),
    ("eval_bn", 2, 64, 64, 8, 8, 1, 0, "prelu", True, False),
]


@pytest.mark.parametrize("case", BN_CASES, ids=[c[0] for c in BN_CASES])
def test_conv_bn_act(dev, case):
    F = P("functional")
    name, n, cin, cout, h, w, stride, pmode, actn, has_res, train = case
    act = dict(none=F.ACT_NONE, leaky=F.ACT_LEAKY, prelu=F.ACT_PRELU)[actn]
    slope = 0.2 if actn == "leaky" else 0.25
    x = bfr(filler.tensor("x:" + name, (n, cin, h, w)))
    wt = bfr(filler.tensor("w:" + name, (cout, cin, 3, 3), float(np.sqrt(3.0 / (cin * 9)))))
    b = filler.tensor("b:" + name, (cout,), 0.1)
    gamma = filler.tensor("g:" + name, (cout,), 0.2, 1.0)
    beta = filler.tensor("be:" + name, (cout,), 0.1)
    rm0 = filler.tensor("rm:" + name, (cout,), 0.1)
    rv0 = filler.tensor("rv:" + name, (cout,), 0.3, 1.0)
    # reference
    xr, wr, br = x.clone().requires_grad_(True), wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
    gr, ber = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    ar = torch.tensor([slope], requires_grad=True)
    rm, rv = rm0.clone(), rv0.clone()
    yr = TF.conv2d(pad_ref(xr, 1, pmode), wr, br, stride=stride)
    zr = TF.batch_norm(yr, rm, rv, gr, ber, training=train, momentum=0.1, eps=1e-5)
    zr = act_ref(F, zr, act, ar if act == F.ACT_PRELU else slope)
    res = None
    if has_res:
        res = bfr(filler.tensor("r:" + name, tuple(zr.shape)))
        resr = res.clone().requires_grad_(True)
        zr = zr + resr
    probe = bfr(filler.tensor("p:" + name, tuple(zr.shape)))
    (zr * probe).sum().backward()
    # HIP
    xg = to_nhwc(x).to(dev).requires_grad_(True)
    wg, bg = wt.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    gg, beg = gamma.to(dev).requires_grad_(True), beta.to(dev).requires_grad_(True)
    ag = torch.tensor([slope], device=dev, requires_grad=True) if act == F.ACT_PRELU else None
    rmg, rvg = rm0.to(dev), rv0.to(dev)
    nbt = torch.zeros((), dtype=torch.int64, device=dev)
    resg = to_nhwc(res).to(dev).requires_grad_(True) if has_res else None
    cfg = dict(stride=stride, pad=1, pad_mode=pmode, act=act, slope=slope, train=train)
    zg = F.ConvBNAct.apply(xg, wg, bg, gg, beg, rmg, rvg, nbt, ag, resg, cfg)
    zg.backward(to_nhwc(probe).to(dev))
    torch.cuda.synchronize()
    assert rel_err(from_nhwc(zg.detach().cpu(), cout), zr.detach()) < 2e-2
    if train:
        assert rel_err(rmg.cpu(), rm) < 2e-3 and rel_err(rvg.cpu(), rv) < 2e-3 and int(nbt) == 1
    else:
        assert torch.equal(rmg.cpu(), rm0) and int(nbt) == 0
    tol = 4e-2
    assert rel_err(from_nhwc(xg.grad.cpu(), cin), xr.grad) < tol, ("dx", rel_err(from_nhwc(xg.grad.cpu(), cin), xr.grad))
    assert rel_err(wg.grad.cpu(), wr.grad) < tol, ("dw", rel_err(wg.grad.cpu(), wr.grad))
    assert rel_err(gg.grad.cpu(), gr.grad) < tol, ("dgamma", rel_err(gg.grad.cpu(), gr.grad))
    assert rel_err(beg.grad.cpu(), ber.grad) < tol, ("dbeta", rel_err(beg.grad.cpu(), ber.grad))
    if ag is not None:
        assert rel_err(ag.grad.cpu(), ar.grad) < tol
    if has_res:
        assert rel_err(from_nhwc(resg.grad.cpu(), cout), resr.grad) < 1e-6
    if not train:
        assert rel_err(bg.grad.cpu(), br.grad) < tol
    else:
        assert bg.grad is None      # analytically zero in front of a train-mode BatchNorm: not produced (functional.ConvBNAct)


@pytest.mark.parametrize("actn,cout", [("tanh", 3), ("sigmoid", 3), ("none", 5)])
def test_conv_out_nchw(dev, actn, cout):
    F = P("functional")
    act = dict(tanh=F.ACT_TANH, sigmoid=F.ACT_SIGMOID, none=F.ACT_NONE)[actn]
    k, pad = (9, 4) if actn == "tanh" else (1, 0)
    x = bfr(filler.tensor("x:o" + actn, (2, 64, 9, 11)))
    wt = bfr(filler.tensor("w:o" + actn, (cout, 64, k, k), float(np.sqrt(3.0 / (64 * k * k)))))
    b = filler.tensor("b:o" + actn, (cout,), 0.1)
    xr, wr, br = x.clone().requires_grad_(True), wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = act_ref(F, TF.conv2d(xr, wr, br, padding=pad), act, 0.0)
    probe = filler.tensor("p:o" + actn, tuple(yr.shape))
    (yr * probe).sum().backward()
    xg = to_nhwc(x).to(dev).requires_grad_(True)
    wg, bg = wt.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    yg = F.ConvOutNCHW.apply(xg, wg, bg, dict(stride=1, pad=pad, act=act))
    assert yg.dtype == torch.float32 and tuple(yg.shape) == tuple(yr.shape)
    yg.backward(probe.to(dev))
    torch.cuda.synchronize()
    assert rel_err(yg.detach().cpu(), yr.detach()) < 2e-5      # fp32 output of fp32 accumulation
    assert rel_err(from_nhwc(xg.grad.cpu(), 64), xr.grad) < 2.5e-2
    assert rel_err(wg.grad.cpu(), wr.grad) < 2.5e-2
    assert rel_err(bg.grad.cpu(), br.grad) < 2.5e-2


def test_layout_roundtrip(dev):
    F = P("functional")
    x = filler.tensor("x:lay", (2, 5, 6, 7)).requires_grad_(True)
    xg = x.detach().to(dev).requires_grad_(True)
    y = F.ToNHWC.apply(xg, torch.bfloat16)
    assert tuple(y.shape) == (2, 6, 7, 8)
    assert torch.equal(y[..., :5].cpu().float(), bfr(x.detach()).permute(0, 2, 3, 1))
    assert float(y[..., 5:].float().abs().max()) == 0.0
    z = F.ToNCHW.apply(y, 5)
    assert torch.equal(z.cpu(), bfr(x.detach()))
    z.backward(torch.ones_like(z))
    assert torch.equal(xg.grad.cpu(), torch.ones(2, 5, 6, 7))


def test_losses_and_adam(dev):
    F = P("functional")
    optim = P("optim")
    a = filler.tensor("l:a", (3, 3, 8, 8))
    b = filler.tensor("l:b", (3, 3, 8, 8))
    for mode, fn in ((0, lambda p, q: (p - q).abs().mean()), (1, lambda p, q: ((p - q) ** 2).mean())):
        ar = a.clone().requires_grad_(True)
        lr_ = fn(ar, b)
        lr_.backward()
        ag = a.to(dev).requires_grad_(True)
        lg = F.DiffLoss.apply(ag, b.to(dev), mode)
        lg.backward()
        assert abs(lg.item() - lr_.item()) < 1e-6 * max(1, abs(lr_.item()))
        assert rel_err(ag.grad.cpu(), ar.grad) < 1e-6
    p = (filler.tensor("l:p", (7, 1), 0.45, 0.5)).clamp(1e-4, 1 - 1e-4)
    p[0, 0] = 0.0          # exercises the -100 clamp of nn.BCELoss
    p[1, 0] = 1.0
    for target in (0.0, 1.0):
        pr = p.clone().requires_grad_(True)
        lref = TF.binary_cross_entropy(pr, torch.full_like(pr, target))
        lref.backward()
        pg = p.to(dev).requires_grad_(True)
        lg = F.bce_const(pg, target)
        lg.backward()
        assert abs(lg.item() - lref.item()) < 1e-5 * max(1, abs(lref.item())), (lg.item(), lref.item())
        fin = torch.isfinite(pr.grad)
        assert rel_err(pg.grad.cpu()[fin], pr.grad[fin]) < 1e-5
    # Adam: 3 steps against torch.optim.Adam
    w0 = filler.tensor("adam:w", (37, 5))
    wr = w0.clone().requires_grad_(True)
    wg = w0.to(dev).requires_grad_(True)
    oref = torch.optim.Adam([wr], lr=1e-2)
    og = optim.FusedAdam([wg], lr=1e-2)
    for it in range(3):
        g = filler.tensor(f"adam:g{it}", (37, 5))
        wr.grad = g.clone()
        wg.grad = g.to(dev)
        oref.step()
        og.step()
    torch.cuda.synchronize()
    assert rel_err(wg.detach().cpu(), wr.detach()) < 1e-6


def test_adam_multi_tensor_matches_per_tensor(dev):
    """dsr_pw_adam_multi (one launch per 64 tensors) matches dsr_pw_adam to the last ulp or two, for 150 ragged tensors and
    against torch.optim.Adam (train_GAN.py:35-36)."""
    optim = P("optim")
    shapes = [(1 + (7 * i) % 33, 1 + (5 * i) % 19) for i in range(149)] + [(9000,)]
    w0 = [filler.tensor(f"am:w{i}", s) for i, s in enumerate(shapes)]
    a = [w.to(dev).requires_grad_(True) for w in w0]
    b = [w.to(dev).requires_grad_(True) for w in w0]
    r = [w.clone().requires_grad_(True) for w in w0]
    oa, ob = optim.FusedAdam(a, lr=3e-3), optim.FusedAdam(b, lr=3e-3)
    ob.MULTI_MAX = 0                      # per-tensor launches
    oref = torch.optim.Adam(r, lr=3e-3)
    for it in range(3):
        for i, s in enumerate(shapes):
            g = filler.tensor(f"am:g{it}:{i}", s)
            a[i].grad, b[i].grad, r[i].grad = g.to(dev), g.to(dev), g.clone()
        oa.step()
        ob.step()
        oref.step()
    torch.cuda.synchronize()
    for i in range(len(shapes)):
        assert rel_err(a[i].detach().cpu(), b[i].detach().cpu()) < 1e-6, i      # FMA contraction may differ by an ulp
        assert rel_err(a[i].detach().cpu(), r[i].detach()) < 1e-6, i


def test_adam_keeps_the_dense_shadow_current(dev):
    """The bf16 image of the dense head's matrix (functional.shadow16) is rewritten by the Adam launch itself and must
    be recognised as current afterwards: no second cast pass per step, and never a stale image."""
    F, optim = P("functional"), P("optim")
    w = filler.tensor("sh:w", (24, 136)).to(dev).requires_grad_(True)
    sh0 = F.shadow16(w, torch.bfloat16)
    assert torch.equal(sh0, w.detach().to(torch.bfloat16))
    opt = optim.FusedAdam([w], lr=1e-2)
    for it in range(2):
        w.grad = filler.tensor(f"sh:g{it}", (24, 136)).to(dev)
        opt.step()
        key = (id(w), torch.bfloat16, tuple(w.shape))
        assert F._shadow_cache[key][0] == F._version(w)                 # marked current: shadow16 will not re-cast
        sh = F.shadow16(w, torch.bfloat16)
        assert sh.data_ptr() == sh0.data_ptr()
        assert torch.equal(sh, w.detach().to(torch.bfloat16))
    with torch.no_grad():                                               # an outside in-place edit must invalidate it
        w.mul_(0.5)
    assert torch.equal(F.shadow16(w, torch.bfloat16), w.detach().to(torch.bfloat16))


PERSIST_CASES = [  # name, N, H, W, Cin, Cout, stride, act, stats, pixel_shuffle
    ("fwd_128_stats", 4, 192, 192, 64, 128, 1, 0, True, False),
    ("fwd_128_leaky", 4, 192, 192, 64, 128, 1, 1, False, False),
    ("fwd_256_pixshuf", 4, 192, 192, 64, 256, 1, 2, False, True),
    ("fwd_64_s2", 6, 400, 368, 64, 64, 2, 1, True, False),          # persistent 128x64 forward (stats) + 4 dgrad classes
    ("fwd_64_s1_relu", 3, 272, 260, 128, 64, 1, 3, False, False),    # persistent forward, channel-major C tile
    ("dgrad_256_64", 4, 192, 192, 64, 256, 1, 0, False, False),      # dgrad of the PixelShuffle conv: 36 K-steps, N = 64
    ("fwd_192_ragged", 5, 190, 170, 128, 192, 1, 3, False, False),
]


@pytest.mark.parametrize("case", PERSIST_CASES, ids=[c[0] for c in PERSIST_CASES])
def test_persistent_conv_equals_per_image_launches(dev, case):
    """64-wide GEMMs with at least two tiles per resident block run on conv_gemm_persist_kernel (tile loop, C tile
    in the last stage, counted vmcnt across the tile boundary); the same convolution issued one image at a time stays
    on the one-tile-per-block kernel.  Both must agree BIT FOR BIT (forward, statistics sums, input gradient); the
    128-wide cases check the same batch-split invariance on the one-tile kernel."""
    import ctypes as C
    L = P("_lib")
    lib = L.lib()
    _, n, h, w, cin, cout, stride, act, stats, ps = case
    ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    torch.manual_seed(7)
    x = (torch.rand(n, h, w, cin, device=dev) - 0.5).to(torch.bfloat16)
    wt = (torch.rand(cout, cin, 3, 3, device=dev) - 0.5) * 0.2
    bias = torch.rand(cout, device=dev) - 0.5
    prelu = torch.tensor([0.2], device=dev)

    def run(xb):
        nb = xb.shape[0]
        d = L.ConvDesc(L.BF16, nb, h, w, cin, cout, 3, 3, stride, 1, 0)
        oh, ow = (h + 2 - 3) // stride + 1, (w + 2 - 3) // stride + 1
        wf = torch.empty(lib.dsr_conv_packed_elems(C.byref(d), 0), dtype=torch.bfloat16, device=dev)
        wd = torch.empty(lib.dsr_conv_packed_elems(C.byref(d), 1), dtype=torch.bfloat16, device=dev)
        L.check(lib.dsr_conv_pack_weight(C.byref(d), ptr(wt), ptr(wf), ptr(wd), st))
        if ps:
            y = torch.full((nb, 2 * oh, 2 * ow, cout // 4), float("nan"), dtype=torch.bfloat16, device=dev)
        else:
            y = torch.full((nb, oh, ow, cout), float("nan"), dtype=torch.bfloat16, device=dev)
        rows = lib.dsr_conv_stats_rows(C.byref(d))
        part = torch.zeros((rows + 64) * 2 * cout, dtype=torch.float32, device=dev) if stats else None
        ep = L.Epilogue(act, 0.2, ptr(prelu) if act == 2 else None, ptr(bias), ptr(part), int(ps), None)
        L.check(lib.dsr_conv_fwd(C.byref(d), ptr(xb), ptr(wf), C.byref(ep), ptr(y), st))
        ssum = part[: rows * 2 * cout].view(rows, 2, cout).double().sum(0) if stats else None
        dx = None
        if not ps:
            dy = y.clone()          # any finite tensor of the output shape
            dx = torch.full_like(xb, float("nan"))
            wsz = lib.dsr_conv_dgrad_workspace(C.byref(d))
            ws = torch.empty(max(wsz, 16), dtype=torch.uint8, device=dev)
            L.check(lib.dsr_conv_dgrad(C.byref(d), ptr(dy), ptr(wd), ptr(dx), ptr(ws), wsz, st))
        return y, ssum, dx

    y_all, s_all, dx_all = run(x)
    ys, ss, dxs = zip(*[run(x[i:i + 1].contiguous()) for i in range(n)])
    torch.cuda.synchronize()
    assert torch.isfinite(y_all.float()).all()
    assert torch.equal(y_all, torch.cat(ys))
    if stats:
        ref = sum(ss)
        assert (s_all - ref).abs().max().item() <= 1e-6 * ref.abs().max().item()
    if dx_all is not None:
        assert torch.isfinite(dx_all.float()).all()
        assert torch.equal(dx_all, torch.cat(dxs))


@pytest.mark.parametrize("n,h,w,actn", [(3, 37, 70, "leaky"), (2, 130, 200, "relu"), (1, 16, 24, "none"), (4, 64, 64, "leaky")])
def test_first_layer_fused_backward(dev, n, h, w, actn):
    """An image input needs no gradient (discriminator.py:22 on HR patches): ConvAct.backward then runs
    dsr_conv_first_bwd_recompute -- activation mask + bias gradient + weight gradient in one pass, the mask taken from the
    pre-activation it recomputes from the image and the layer's weights (the stored activation is not read) -- or, with
    functional.FIRST_BWD_RECOMPUTE off, dsr_conv_first_bwd, which reads the stored activation.  Both must match the fp32
    reference, each other (the recomputed sign can differ from the stored one only where the pre-activation is within fp32
    summation noise of zero), and the unfused path (act_bwd + wgrad) that the layer takes when its input requires grad."""
    F = P("functional")
    act = dict(none=F.ACT_NONE, leaky=F.ACT_LEAKY, relu=F.ACT_RELU)[actn]
    x = bfr(filler.tensor(f"fl:x{h}", (n, 3, h, w)))
    wt = bfr(filler.tensor(f"fl:w{h}", (64, 3, 3, 3), float(np.sqrt(3.0 / 27))))
    b = filler.tensor(f"fl:b{h}", (64,), 0.1)
    wr, br = wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = act_ref(F, TF.conv2d(x, wr, br, padding=1), act, 0.2)
    probe = bfr(filler.tensor(f"fl:p{h}", tuple(yr.shape)))
    (yr * probe).sum().backward()
    cfg = dict(stride=1, pad=1, pad_mode=0, act=act, slope=0.2)
    grads = []
    try:
        for need_dx, recompute in ((False, True), (False, False), (True, True)):
            F.FIRST_BWD_RECOMPUTE = recompute
            xg = to_nhwc(x).to(dev).requires_grad_(need_dx)
            wg, bg = wt.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
            yg = F.ConvAct.apply(xg, wg, bg, None, cfg)
            yg.backward(to_nhwc(probe, 64).to(dev))
            torch.cuda.synchronize()
            grads.append((wg.grad.cpu(), bg.grad.cpu()))
    finally:
        F.FIRST_BWD_RECOMPUTE = True
    (dw_f, db_f), (dw_s, db_s), (dw_u, db_u) = grads
    assert rel_err(dw_f, wr.grad) < 2.5e-2 and rel_err(db_f, br.grad) < 2.5e-2
    assert rel_err(dw_s, wr.grad) < 2.5e-2 and rel_err(db_s, br.grad) < 2.5e-2
    assert rel_err(dw_f, dw_s) < 1e-3 and rel_err(db_f, db_s) < 1e-3          # recomputed sign vs stored activation
    # fused vs unfused: the fused kernel multiplies in fp32 and rounds g once, the unfused path stores g in bf16 first
    assert rel_err(dw_f, dw_u) < 5e-3 and rel_err(db_f, db_u) < 5e-3


@pytest.mark.parametrize("n,h,w,slope", [(2, 32, 48, 0.25), (1, 70, 130, 0.1), (3, 16, 16, 0.25), (1, 256, 64, 0.3)])
def test_tail_input_gradient_with_pixel_shuffle_prelu_backward(dev, n, h, w, slope):
    """dsr_conv_dgrad_ps -- the input gradient of the 9x9 64 -> 3 tail (generator.py:78) with the backward of the
    PixelShuffle(2) + PReLU in front of it (generator.py:37-39) in its epilogue: masked, UN-shuffled gradient of the shuffle
    conv's output, its column sums (bias gradient) and the PReLU-weight gradient, without ever writing the 64-channel gradient
    at the high resolution -- against the two launches it replaces (dsr_conv_dgrad, then dsr_pw_act_bwd with pixshuf = 1, which
    sees the gradient rounded to bf16 first) and against float64 PyTorch on the same bf16 operands.  Strips with ragged right
    edges (width 48, 130), several row bands per strip (height 256), an odd number of rows per band before rounding."""
    import ctypes as C
    L = P("_lib")
    F = P("functional")
    lib = L.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device="cpu").manual_seed(h + w)
    dy3 = bfr(torch.rand(n, 3, h, w, generator=g) - 0.5)
    wt = bfr((torch.rand(3, 64, 9, 9, generator=g) - 0.5) * 0.05)
    out = bfr(torch.randn(n, 64, h, w, generator=g))             # the activation output: sign = PReLU branch
    d = L.ConvDesc(L.BF16, n, h, w, 64, 3, 9, 9, 1, 4, 0)
    assert lib.dsr_conv_dgrad_ps_supported(C.byref(d)) == 1
    wf = torch.empty(lib.dsr_conv_packed_elems(C.byref(d), 0), dtype=torch.bfloat16, device=dev)
    wd = torch.empty(lib.dsr_conv_packed_elems(C.byref(d), 1), dtype=torch.bfloat16, device=dev)
    L.check(lib.dsr_conv_pack_weight(C.byref(d), wt.to(dev).data_ptr(), wf.data_ptr(), wd.data_ptr(), st))
    dyg, outg = to_nhwc(dy3).to(dev), to_nhwc(out, 64).to(dev)
    prelu = torch.full((1,), slope, device=dev)
    scr = lib.dsr_pw_scratch_rows()
    lh, lw = h // 2, w // 2
    # ---- one launch
    rows = lib.dsr_conv_dgrad_ps_rows(C.byref(d))
    part_f = torch.full(((rows + scr) * 2 * 256,), float("nan"), dtype=torch.float32, device=dev)
    dyu_f = torch.full((n, lh, lw, 256), float("nan"), dtype=torch.bfloat16, device=dev)
    L.check(lib.dsr_conv_dgrad_ps(C.byref(d), dyg.data_ptr(), wd.data_ptr(), outg.data_ptr(), prelu.data_ptr(), dyu_f.data_ptr(),
                                  part_f.data_ptr(), st))
    # ---- two launches
    dx = torch.empty((n, h, w, 64), dtype=torch.bfloat16, device=dev)
    wsz = lib.dsr_conv_dgrad_workspace(C.byref(d))
    ws = torch.empty(max(wsz, 16), dtype=torch.uint8, device=dev)
    L.check(lib.dsr_conv_dgrad(C.byref(d), dyg.data_ptr(), wd.data_ptr(), dx.data_ptr(), ws.data_ptr(), wsz, st))
    p = n * lh * lw
    blocks, rpb = F._reduce_blocks(p)
    part_t = torch.empty((blocks + scr) * 2 * 256, dtype=torch.float32, device=dev)
    dyu_t = torch.empty((n, lh, lw, 256), dtype=torch.bfloat16, device=dev)
    L.check(lib.dsr_pw_act_bwd(L.BF16, dx.data_ptr(), outg.data_ptr(), dyu_t.data_ptr(), n, lh, lw, 256, 64, 1, F.ACT_PRELU, 0.0,
                               prelu.data_ptr(), blocks, rpb, part_t.data_ptr(), st))
    torch.cuda.synchronize()
    # ---- float64 reference: d = conv_transpose (= dgrad), g = d * PReLU'(out), un-shuffle, sums
    dref = TF.conv_transpose2d(dy3.double(), wt.double(), padding=4)                      # [n, 64, h, w]
    od = out.double()
    gref = torch.where(od >= 0, dref, dref * slope)
    dyu_ref = TF.pixel_unshuffle(gref, 2)                                                  # channel 4c + 2i + j
    db_ref = dyu_ref.sum(dim=(0, 2, 3))
    dp_ref = float((dref * (od / slope) * (od < 0)).sum())
    got_f, got_t = from_nhwc(dyu_f.cpu(), 256).double(), from_nhwc(dyu_t.cpu(), 256).double()
    assert torch.isfinite(got_f).all()
    scale = float(dyu_ref.abs().max())
    assert float((got_f - dyu_ref).abs().max()) < 6e-3 * scale                            # one bf16 rounding of g
    assert float((got_t - dyu_ref).abs().max()) < 1.2e-2 * scale                          # two (d, then g)
    for part, nb, tol in ((part_f, rows, 2e-3), (part_t, blocks, 4e-3)):
        pr = part[:nb * 2 * 256].view(nb, 2, 256).cpu().double().sum(0)
        assert float((pr[0] - db_ref).abs().max()) < tol * float(dyu_ref.abs().sum(dim=(0, 2, 3)).max())
        assert abs(float(pr[1].sum()) - dp_ref) < tol * float((dref * (od / slope) * (od < 0)).abs().sum())


@pytest.mark.parametrize("n,h,w,cin,cout,actn", [(2, 32, 32, 128, 128, "leaky"), (3, 64, 64, 256, 256, "leaky"), (1, 4, 512, 64, 128, "none"),
                                                  (40, 64, 64, 128, 64, "leaky"), (2, 32, 32, 512, 512, "leaky")])
def test_conv_dgrad_with_batchnorm_backward_sums(dev, n, h, w, cin, cout, actn):
    """dsr_conv_dgrad_bn -- the input gradient of a 3x3 stride-2 layer (discriminator.py:31,33,35) that also forms the two sums
    the BatchNorm backward of the layer in front needs (sum g, sum g*y with g = dx * LeakyReLU'(scale*y + shift)) while dx is
    on its way out -- against dsr_conv_dgrad (dx must agree BIT FOR BIT: the same kernel body), against the separate reduce
    pass it replaces (dsr_pw_bn_act_bwd_reduce -> dsr_pw_bn_bwd_finalize: dgamma, dbeta, c1, c2 from both sets of partial rows)
    and against float64 sums over the stored dx.  One to eight 64-channel slices, tiles of 16 rows down to a row segment,
    more tiles than persistent blocks (batch 40: 320 tiles on 256 blocks, blocks with one and with two tiles)."""
    import ctypes as C
    L = P("_lib")
    F = P("functional")
    lib = L.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    act = dict(leaky=F.ACT_LEAKY, none=F.ACT_NONE)[actn]
    g = torch.Generator(device="cpu").manual_seed(cin + h)
    oh, ow = h // 2, w // 2
    dy = bfr(torch.rand(n, cout, oh, ow, generator=g) - 0.5)
    wt = bfr((torch.rand(cout, cin, 3, 3, generator=g) - 0.5) * 0.1)
    y = bfr(torch.randn(n, cin, h, w, generator=g))
    scale = torch.rand(cin, generator=g) + 0.5
    scale[::3] *= -1
    shift = (torch.rand(cin, generator=g) - 0.5)
    mean = torch.rand(cin, generator=g) - 0.5
    rstd = torch.rand(cin, generator=g) + 0.5
    d = L.ConvDesc(L.BF16, n, h, w, cin, cout, 3, 3, 2, 1, 0)
    assert lib.dsr_conv_dgrad_bn_supported(C.byref(d)) == 1
    wf = torch.empty(lib.dsr_conv_packed_elems(C.byref(d), 0), dtype=torch.bfloat16, device=dev)
    wd = torch.empty(lib.dsr_conv_packed_elems(C.byref(d), 1), dtype=torch.bfloat16, device=dev)
    L.check(lib.dsr_conv_pack_weight(C.byref(d), wt.to(dev).data_ptr(), wf.data_ptr(), wd.data_ptr(), st))
    dyg, yg = to_nhwc(dy, cout).to(dev), to_nhwc(y, cin).to(dev)
    scg, shg, mg, rg = scale.to(dev), shift.to(dev), mean.to(dev), rstd.to(dev)
    scr = lib.dsr_pw_scratch_rows()
    # ---- one launch
    rows = lib.dsr_conv_dgrad_bn_rows(C.byref(d))
    part = torch.full(((rows + scr) * 3 * cin,), float("nan"), dtype=torch.float32, device=dev)
    dx_f = torch.full((n, h, w, cin), float("nan"), dtype=torch.bfloat16, device=dev)
    L.check(lib.dsr_conv_dgrad_bn(C.byref(d), dyg.data_ptr(), wd.data_ptr(), dx_f.data_ptr(), yg.data_ptr(), scg.data_ptr(),
                                  shg.data_ptr(), act, 0.2, part.data_ptr(), st))
    # ---- dgrad, then the reduce pass
    dx_t = torch.full((n, h, w, cin), float("nan"), dtype=torch.bfloat16, device=dev)
    wsz = lib.dsr_conv_dgrad_workspace(C.byref(d))
    ws = torch.empty(max(wsz, 16), dtype=torch.uint8, device=dev)
    L.check(lib.dsr_conv_dgrad(C.byref(d), dyg.data_ptr(), wd.data_ptr(), dx_t.data_ptr(), ws.data_ptr(), wsz, st))
    p = n * h * w
    blocks, rpb = F._bn_bwd_blocks(p, act)
    part_t = torch.empty((blocks + scr) * 3 * cin, dtype=torch.float32, device=dev)
    L.check(lib.dsr_pw_bn_act_bwd_reduce(L.BF16, dx_t.data_ptr(), yg.data_ptr(), scg.data_ptr(), shg.data_ptr(), mg.data_ptr(),
                                         rg.data_ptr(), p, cin, blocks, rpb, act, 0.2, None, part_t.data_ptr(), st))
    outs = []
    for pt, nb in ((part, rows), (part_t, blocks)):
        o = [torch.empty(cin, dtype=torch.float32, device=dev) for _ in range(4)]
        L.check(lib.dsr_pw_bn_bwd_finalize(pt.data_ptr(), nb, cin, cin, float(p), mg.data_ptr(), rg.data_ptr(), o[0].data_ptr(),
                                           o[1].data_ptr(), None, o[2].data_ptr(), o[3].data_ptr(), st))
        outs.append([t.cpu().double() for t in o])
    torch.cuda.synchronize()
    assert torch.equal(dx_f, dx_t)
    rowsum = part[:rows * 3 * cin].view(rows, 3, cin).cpu().double().sum(0)
    assert torch.isfinite(rowsum).all() and float(rowsum[2].abs().max()) == 0.0
    dxd, yd = from_nhwc(dx_t.cpu(), cin).double(), y.double()
    z = (y * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))                 # fp32, as the kernels form it
    gref = dxd * (torch.where(z < 0, 0.2, 1.0).double() if actn == "leaky" else 1.0)
    sg_ref, sgy_ref = gref.sum(dim=(0, 2, 3)), (gref * yd).sum(dim=(0, 2, 3))
    tol_g, tol_gy = 2e-5 * float(gref.abs().sum(dim=(0, 2, 3)).max()), 2e-5 * float((gref * yd).abs().sum(dim=(0, 2, 3)).max())
    assert float((rowsum[0] - sg_ref).abs().max()) < tol_g and float((rowsum[1] - sgy_ref).abs().max()) < tol_gy
    for a_, b_ in zip(*outs):                       # dgamma, dbeta, c1, c2: the two sets of partial rows through the same finalize
        assert float((a_ - b_).abs().max()) < 1e-4 * float(b_.abs().max()) + 1e-6


@pytest.mark.parametrize("n,h,w,dt", [(2, 40, 72, "bf16"), (1, 128, 128, "bf16"), (1, 19, 35, "f16")])
def test_generator_head_9x9_one_kernel_row_per_k_step(dev, n, h, w, dt, monkeypatch):
    """conv_rgb9_kernel -- generator.py:48, Conv2d(3, 64, 9, 1, 4) + PReLU, with the RGB halo packed to 3 channels in LDS so
    that one kernel row (9 taps x 3 channels) is one MFMA k-step read by a 2-byte-aligned 16-byte LDS load -- against an fp64
    conv2d on the same rounded operands (one output rounding) and against the gather kernel it replaces (DSR_CONV_RGB9=0:
    81 taps of 8 stored channels).  The k slots 27..31 of a fragment lie over the pixels x + 5, x + 6 of the halo row, outside
    the window: an Inf planted in the image must reach exactly the 9 x 9 outputs whose window holds it, no column further."""
    import ctypes as C
    L = P("_lib")
    F = P("functional")
    lib = L.lib()
    tdt, code = (torch.bfloat16, L.BF16) if dt == "bf16" else (torch.float16, L.F16)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device="cpu").manual_seed(h * w)
    x = (torch.rand(n, 3, h, w, generator=g)).to(tdt).float()
    wt = ((torch.rand(64, 3, 9, 9, generator=g) - 0.5) * 0.2).to(tdt).float()
    b = (torch.rand(64, generator=g) - 0.5) * 0.2
    d = L.ConvDesc(code, n, h, w, 3, 64, 9, 9, 1, 4, 0)
    wf = torch.empty(lib.dsr_conv_packed_elems(C.byref(d), 0), dtype=tdt, device=dev)
    wd = torch.empty(max(lib.dsr_conv_packed_elems(C.byref(d), 1), 8), dtype=tdt, device=dev)
    L.check(lib.dsr_conv_pack_weight(C.byref(d), wt.to(dev).data_ptr(), wf.data_ptr(), wd.data_ptr(), st))
    bg, ag = b.to(dev), torch.tensor([0.25], device=dev)
    ep = L.Epilogue(F.ACT_PRELU, 0.0, ag.data_ptr(), bg.data_ptr(), None, 0, None, None, None, None)

    def run(xin, rgb9):
        monkeypatch.setenv("DSR_CONV_RGB9", "1" if rgb9 else "0")
        assert lib.dsr_conv_kernel_name(C.byref(d), 0, C.byref(ep)).decode().startswith("conv_rgb9" if rgb9 else "conv_gemm")
        xg = torch.zeros(n, h, w, 8, dtype=tdt, device=dev)
        xg[..., :3] = xin.permute(0, 2, 3, 1).to(tdt).to(dev)
        y = torch.full((n, h, w, 64), float("nan"), dtype=tdt, device=dev)
        L.check(lib.dsr_conv_fwd(C.byref(d), xg.data_ptr(), wf.data_ptr(), C.byref(ep), y.data_ptr(), st))
        torch.cuda.synchronize()
        return y.float().cpu().permute(0, 3, 1, 2)
    y_new, y_old = run(x, True), run(x, False)
    ref = TF.conv2d(x.double(), wt.double(), b.double(), padding=4)
    ref = torch.where(ref >= 0, ref, 0.25 * ref)
    ulp = 2.0 ** -8 if dt == "bf16" else 2.0 ** -11
    tol = ulp * ref.abs().clamp_min(0.05) + 2e-5 * 243           # one output rounding + fp32 accumulation of 243 products
    assert torch.isfinite(y_new).all()
    assert float(((y_new - ref).abs() / tol).max()) <= 1.0, float(((y_new - ref).abs() / tol).max())
    assert float(((y_new - y_old).abs() / (2 * tol)).max()) <= 1.0
    # ---- weight gradient (conv_rgb9_wgrad_kernel: one GEMM per (tile row, kernel row) over the row's 32 pixels) against float64
    # and against the tap-per-MFMA kernel it replaces
    dy = ((torch.rand(n, 64, h, w, generator=g) - 0.5)).to(tdt).float()
    dyg = dy.permute(0, 2, 3, 1).contiguous().to(tdt).to(dev)
    xg8 = torch.zeros(n, h, w, 8, dtype=tdt, device=dev)
    xg8[..., :3] = x.permute(0, 2, 3, 1).to(tdt).to(dev)

    def wgrad(rgb9):
        monkeypatch.setenv("DSR_CONV_RGB9", "1" if rgb9 else "0")
        assert lib.dsr_conv_kernel_name(C.byref(d), 2, None).decode() == ("conv_rgb9_wgrad_kernel" if rgb9 else "conv_wgrad_taps_kernel")
        wsz = lib.dsr_conv_wgrad_workspace(C.byref(d))
        ws = torch.full((wsz // 4,), float("nan"), dtype=torch.float32, device=dev)
        dw = torch.full((64, 3, 9, 9), float("nan"), dtype=torch.float32, device=dev)
        L.check(lib.dsr_conv_wgrad(C.byref(d), xg8.data_ptr(), dyg.data_ptr(), dw.data_ptr(), ws.data_ptr(), wsz, st))
        torch.cuda.synchronize()
        return dw.cpu().double()
    dw_new, dw_old = wgrad(True), wgrad(False)
    xr = x.double().requires_grad_(True)
    wr = wt.double().requires_grad_(True)
    TF.conv2d(xr, wr, None, padding=4).backward(dy.double())
    scale_w = float(wr.grad.abs().max())
    assert torch.isfinite(dw_new).all()
    assert float((dw_new - wr.grad).abs().max()) <= 2e-5 * scale_w * 8, float((dw_new - wr.grad).abs().max()) / scale_w
    assert float((dw_new - dw_old).abs().max()) <= 4e-5 * scale_w * 8
    # an Inf at (iy, ix): outputs within 4 pixels of it are non-finite (Inf, or NaN where the weight is 0 or signs cancel),
    # every other output is what it was
    iy, ix = h // 2, min(w - 1, 37)
    xi = x.clone()
    xi[0, 1, iy, ix] = float("inf")
    y_inf = run(xi, True)
    hit = torch.zeros(n, 1, h, w, dtype=torch.bool)
    hit[0, 0, max(0, iy - 4):iy + 5, max(0, ix - 4):ix + 5] = True
    bad = ~torch.isfinite(y_inf)
    assert not bool((bad & ~hit).any()), "a non-finite input reached an output whose window does not hold it"
    assert bool(bad[0, :, iy, ix].any())
    assert torch.equal(torch.where(hit, torch.zeros(()), y_inf), torch.where(hit, torch.zeros(()), y_new))


@pytest.mark.parametrize("n,h,w,cin,actn", [(2, 6, 512, 3, "leaky"), (1, 4, 1024, 3, "relu"), (3, 2, 512, 1, "leaky"), (130, 4, 512, 3, "leaky")])
def test_first_two_layers_fused_backward(dev, n, h, w, cin, actn):
    """dsr_conv_dgrad_first_bwd -- the input gradient of discriminator.py:29 (Conv2d(64,64,3,2,1)) and the whole backward of
    :25-27 (Conv2d(3,64,3,1,1) + LeakyReLU: mask, bias gradient, weight gradient) as ONE launch that never writes the gradient
    of the 64-channel activation in between -- against (a) the two launches it replaces (dsr_conv_dgrad, then
    dsr_conv_first_bwd_recompute on the stored gradient, which is rounded to bf16 before the mask where the one launch rounds
    the masked fp32 value once) and (b) float64 PyTorch on the same bf16 operands with g rounded where each path rounds it.  Image rows of one and two tiles,
    the top / bottom / left / right image borders in every tile, a one-channel image, more tiles than persistent blocks."""
    import ctypes as C
    L = P("_lib")
    F = P("functional")
    lib = L.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    act = dict(leaky=F.ACT_LEAKY, relu=F.ACT_RELU)[actn]
    g = torch.Generator(device="cpu").manual_seed(h * w + n)
    x = bfr(torch.rand(n, cin, h, w, generator=g) * 2 - 1)
    w0 = (torch.rand(64, cin, 3, 3, generator=g) - 0.5) * 0.6
    b0 = (torch.rand(64, generator=g) - 0.5) * 0.2
    w1 = bfr((torch.rand(64, 64, 3, 3, generator=g) - 0.5) * 0.1)
    oh, ow = h // 2, w // 2
    dy = bfr(torch.rand(n, 64, oh, ow, generator=g) - 0.5)
    d0 = L.ConvDesc(L.BF16, n, h, w, cin, 64, 3, 3, 1, 1, 0)
    d1 = L.ConvDesc(L.BF16, n, h, w, 64, 64, 3, 3, 2, 1, 0)
    assert lib.dsr_conv_dgrad_first_bwd_supported(C.byref(d0), C.byref(d1), act) == 1
    wd1 = torch.empty(lib.dsr_conv_packed_elems(C.byref(d1), 1), dtype=torch.bfloat16, device=dev)
    wf1 = torch.empty(lib.dsr_conv_packed_elems(C.byref(d1), 0), dtype=torch.bfloat16, device=dev)
    L.check(lib.dsr_conv_pack_weight(C.byref(d1), w1.to(dev).data_ptr(), wf1.data_ptr(), wd1.data_ptr(), st))
    xg = to_nhwc(x).to(dev)
    dyg = to_nhwc(dy, 64).to(dev)
    w0g, b0g = w0.to(dev), b0.to(dev)
    # ---- one launch
    dw_f = torch.full((64, cin, 3, 3), float("nan"), dtype=torch.float32, device=dev)
    db_f = torch.full((64,), float("nan"), dtype=torch.float32, device=dev)
    wsz = lib.dsr_conv_dgrad_first_bwd_workspace(C.byref(d1))
    ws = torch.empty(wsz, dtype=torch.uint8, device=dev)
    L.check(lib.dsr_conv_dgrad_first_bwd(C.byref(d0), C.byref(d1), dyg.data_ptr(), wd1.data_ptr(), xg.data_ptr(), w0g.data_ptr(),
                                         b0g.data_ptr(), act, 0.2, dw_f.data_ptr(), db_f.data_ptr(), ws.data_ptr(), wsz, st))
    # ---- two launches
    da0 = torch.full((n, h, w, 64), float("nan"), dtype=torch.bfloat16, device=dev)
    wsz2 = lib.dsr_conv_dgrad_workspace(C.byref(d1))
    ws2 = torch.empty(max(wsz2, 16), dtype=torch.uint8, device=dev)
    L.check(lib.dsr_conv_dgrad(C.byref(d1), dyg.data_ptr(), wd1.data_ptr(), da0.data_ptr(), ws2.data_ptr(), wsz2, st))
    dw_t = torch.empty((64, cin, 3, 3), dtype=torch.float32, device=dev)
    db_t = torch.empty((64,), dtype=torch.float32, device=dev)
    wsz3 = lib.dsr_conv_first_bwd_workspace(C.byref(d0))
    ws3 = torch.empty(wsz3, dtype=torch.uint8, device=dev)
    L.check(lib.dsr_conv_first_bwd_recompute(C.byref(d0), xg.data_ptr(), da0.data_ptr(), w0g.data_ptr(), b0g.data_ptr(), act, 0.2,
                                             dw_t.data_ptr(), db_t.data_ptr(), ws3.data_ptr(), wsz3, st))
    torch.cuda.synchronize()
    dw_f, db_f, dw_t, db_t = dw_f.cpu().double(), db_f.cpu().double(), dw_t.cpu().double(), db_t.cpu().double()
    assert torch.isfinite(dw_f).all() and torch.isfinite(db_f).all()
    sw, sb = float(dw_t.abs().max()), float(db_t.abs().max())
    # g rounded once (one launch) against twice (two launches), fp32 partial sums in another order, and a sign that can flip
    # where the recomputed pre-activation is within fp32 noise of zero
    assert float((dw_f - dw_t).abs().max()) < 5e-3 * sw and float((db_f - db_t).abs().max()) < 5e-3 * sb
    # ---- float64 reference
    xd, w0d = x.double(), bfr(w0).double()
    z0 = TF.conv2d(xd, w0d, b0.double(), padding=1)
    da_ref = TF.conv_transpose2d(dy.double(), w1.double(), stride=2, padding=1, output_padding=1)
    mask = torch.where(z0 >= 0, torch.ones_like(z0), torch.full_like(z0, 0.2 if actn == "leaky" else 0.0))
    if actn == "relu":
        mask = (z0 > 0).double()
    for got_w, got_b, twice in ((dw_f, db_f, False), (dw_t, db_t, True)):
        d_in = bfr(da_ref.float()).double() if twice else da_ref          # two launches: the gradient is stored in bf16 first
        gref = bfr((d_in * mask).float()).double()
        dw_r = torch.nn.grad.conv2d_weight(xd, (64, cin, 3, 3), gref, padding=1)
        db_r = gref.sum(dim=(0, 2, 3))
        assert rel_err(got_w, dw_r) < 1e-3 and rel_err(got_b, db_r) < 1e-3, twice


@pytest.mark.parametrize("n,h,w,keep", [(2, 64, 64, True), (3, 37, 70, True), (1, 128, 200, False), (33, 16, 128, True)])
def test_first_two_layers_fused_forward(dev, n, h, w, keep):
    """dsr_conv_first2_fwd -- discriminator.py:25 (Conv2d(3,64,3,1,1) + LeakyReLU(0.2)) and :29 (Conv2d(64,64,3,2,1) in front of its
    BatchNorm) as one launch, the first layer's activation recomputed per tile in LDS -- against the two launches it replaces
    (dsr_conv_fwd twice: the first-layer kernel walks K in the same order, so a0 must agree BIT FOR BIT; the second layer sums
    the same products with its bias added first instead of last) and against plain fp32 PyTorch on the same bf16 operands:
    a0, y1 and the BatchNorm sum / sum of squares of the reference's y1.  Odd sizes (ragged tiles, odd image height: the last
    output row reads one a0 row past... none: zero padding), more tiles than persistent blocks, and a0 = NULL (inference)."""
    import ctypes as C
    L = P("_lib")
    lib = L.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device="cpu").manual_seed(13)
    x = bfr(torch.rand(n, 3, h, w, generator=g) * 2 - 1)
    w0 = bfr((torch.rand(64, 3, 3, 3, generator=g) - 0.5) * 0.6)
    b0 = (torch.rand(64, generator=g) - 0.5) * 0.2
    w1 = bfr((torch.rand(64, 64, 3, 3, generator=g) - 0.5) * 0.1)
    b1 = (torch.rand(64, generator=g) - 0.5) * 0.2
    oh, ow = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    d0 = L.ConvDesc(L.BF16, n, h, w, 3, 64, 3, 3, 1, 1, 0)
    d1 = L.ConvDesc(L.BF16, n, h, w, 64, 64, 3, 3, 2, 1, 0)
    assert lib.dsr_conv_first2_supported(C.byref(d0), C.byref(d1)) == 1

    def pack(d, wt):
        wf = torch.empty(lib.dsr_conv_packed_elems(C.byref(d), 0), dtype=torch.bfloat16, device=dev)
        wd = torch.empty(lib.dsr_conv_packed_elems(C.byref(d), 1), dtype=torch.bfloat16, device=dev)
        L.check(lib.dsr_conv_pack_weight(C.byref(d), wt.to(dev).data_ptr(), wf.data_ptr(), wd.data_ptr(), st))
        return wf
    wf0, wf1 = pack(d0, w0), pack(d1, w1)
    xg = to_nhwc(x).to(dev)
    b0g, b1g = b0.to(dev), b1.to(dev)
    # ---- two launches
    a0_two = torch.full((n, h, w, 64), float("nan"), dtype=torch.bfloat16, device=dev)
    ep0 = L.Epilogue(L.ACT_LEAKY, 0.2, None, b0g.data_ptr(), None, 0, None)
    L.check(lib.dsr_conv_fwd(C.byref(d0), xg.data_ptr(), wf0.data_ptr(), C.byref(ep0), a0_two.data_ptr(), st))
    y1_two = torch.full((n, oh, ow, 64), float("nan"), dtype=torch.bfloat16, device=dev)
    ep1 = L.Epilogue(L.ACT_NONE, 0.0, None, b1g.data_ptr(), None, 0, None)
    L.check(lib.dsr_conv_fwd(C.byref(d1), a0_two.data_ptr(), wf1.data_ptr(), C.byref(ep1), y1_two.data_ptr(), st))
    # ---- one launch
    rows = lib.dsr_conv_first2_stats_rows(C.byref(d0))
    assert 0 < rows <= 256
    a0 = torch.full((n, h, w, 64), float("nan"), dtype=torch.bfloat16, device=dev) if keep else None
    y1 = torch.full((n, oh, ow, 64), float("nan"), dtype=torch.bfloat16, device=dev)
    part = torch.full((rows, 2, 64), float("nan"), dtype=torch.float32, device=dev)
    L.check(lib.dsr_conv_first2_fwd(C.byref(d0), C.byref(d1), xg.data_ptr(), wf0.data_ptr(), b0g.data_ptr(), 0.2, wf1.data_ptr(),
                                    b1g.data_ptr(), a0.data_ptr() if keep else None, y1.data_ptr(), part.data_ptr(), st))
    torch.cuda.synchronize()
    if keep:
        assert torch.equal(a0, a0_two)
    assert torch.isfinite(y1.float()).all() and torch.isfinite(part).all()
    assert rel_err(y1.float(), y1_two.float()) <= 2.0 ** -7
    # ---- fp32 reference
    a0_r = bfr(TF.leaky_relu(TF.conv2d(x, w0, b0, padding=1), 0.2))
    y1_r = TF.conv2d(a0_r, w1, b1, stride=2, padding=1)
    assert rel_err(from_nhwc(a0_two.cpu(), 64), a0_r) <= 1.2e-2
    assert rel_err(from_nhwc(y1.cpu(), 64), y1_r) <= 1.2e-2
    st_ref = torch.stack([y1_r.double().sum((0, 2, 3)), (y1_r.double() ** 2).sum((0, 2, 3))])
    got = part.double().sum(0).cpu()
    assert float((got - st_ref).abs().max() / st_ref.abs().max()) < 2e-3


@pytest.mark.parametrize("b,hw,c,cp,bp", [(5, 72, 100, 128, 8), (64, 1024, 512, 512, 64), (3, 8, 64, 64, 16)])
def test_flatten_tile_kernel_equals_strided_form(dev, b, hw, c, cp, bp):
    """dsr_flatten (NHWC <-> the CHW-flattened operand of the dense head, discriminator.py:37-39,60-62) has a tile form that
    goes through LDS with 16-byte accesses on both sides; DSR_FLATTEN_TILE=0 keeps the strided form.  Pure data movement:
    the two must agree bit for bit in all three modes, and with torch's permute."""
    import ctypes as C
    import os
    L = P("_lib")
    lib = L.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device="cpu").manual_seed(3)
    act = (torch.rand(b, hw, cp, generator=g) - 0.5).to(torch.bfloat16)
    act[..., c:] = 0
    act = act.to(dev)
    flat_src = (torch.rand(b, c * hw, generator=g) - 0.5).to(torch.bfloat16).to(dev)
    res = {}
    old = os.environ.get("DSR_FLATTEN_TILE")
    try:
        for tag in ("1", "0"):
            os.environ["DSR_FLATTEN_TILE"] = tag
            flat = torch.full((b, c * hw), float("nan"), dtype=torch.bfloat16, device=dev)
            L.check(lib.dsr_flatten(L.BF16, act.data_ptr(), flat.data_ptr(), b, hw, c, cp, 0, 0, st))
            flat_t = torch.full((c * hw, bp), float("nan"), dtype=torch.bfloat16, device=dev)
            L.check(lib.dsr_flatten(L.BF16, act.data_ptr(), flat_t.data_ptr(), b, hw, c, cp, bp, 1, st))
            back = torch.full((b, hw, cp), float("nan"), dtype=torch.bfloat16, device=dev)
            L.check(lib.dsr_flatten(L.BF16, flat_src.data_ptr(), back.data_ptr(), b, hw, c, cp, 0, 2, st))
            torch.cuda.synchronize()
            res[tag] = (flat, flat_t, back)
    finally:
        if old is None:
            os.environ.pop("DSR_FLATTEN_TILE", None)
        else:
            os.environ["DSR_FLATTEN_TILE"] = old
    for u, v in zip(res["1"], res["0"]):
        assert torch.isfinite(u.float()).all()
        assert torch.equal(u, v)
    flat, flat_t, back = res["1"]
    ref = act[..., :c].permute(0, 2, 1).reshape(b, c * hw)
    assert torch.equal(flat, ref)
    assert torch.equal(flat_t[:, :b], ref.t()) and float(flat_t[:, b:].float().abs().max() if bp > b else 0.0) == 0.0
    refb = flat_src.reshape(b, c, hw).permute(0, 2, 1)
    assert torch.equal(back[..., :c], refb) and float(back[..., c:].float().abs().max() if cp > c else 0.0) == 0.0


@pytest.mark.parametrize("pmode,stride", [(1, 1), (1, 2), (2, 1)])
def test_padded_coordinate_dma_path_equals_generic_loader(dev, pmode, stride):
    """Reflect (1) / replicate (2) padding with Cin % 64 == 0 runs on the LDS-DMA kernel, which recomputes the padded
    coordinate of every tile row per K-step; DSR_CONV_PADX=0 sends the same launch through the register-staged generic loader.
    Same K order, same tile: the two must agree BIT FOR BIT (models/DIP/utils.py:83-105, pad='reflection')."""
    import ctypes as C
    import os
    L = P("_lib")
    lib = L.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    n, h, w, cin, cout = 2, 21, 30, 128, 160
    d = L.ConvDesc(L.F16, n, h, w, cin, cout, 3, 3, stride, 1, pmode)
    oh, ow = (h + 2 - 3) // stride + 1, (w + 2 - 3) // stride + 1
    g = torch.Generator(device="cpu").manual_seed(11)
    wt = ((torch.rand(cout, cin, 3, 3, generator=g) - 0.5) * 0.1).to(dev)
    wf = torch.empty(lib.dsr_conv_packed_elems(C.byref(d), 0), dtype=torch.float16, device=dev)
    wd = torch.empty(lib.dsr_conv_packed_elems(C.byref(d), 1), dtype=torch.float16, device=dev)
    L.check(lib.dsr_conv_pack_weight(C.byref(d), wt.data_ptr(), wf.data_ptr(), wd.data_ptr(), st))
    x = (torch.rand(n, h, w, cin, generator=g) - 0.5).to(torch.float16).to(dev)
    bias = (torch.rand(cout, generator=g) - 0.5).to(dev)
    outs = []
    old = os.environ.get("DSR_CONV_PADX")
    try:
        for mode in ("0", "1"):
            os.environ["DSR_CONV_PADX"] = mode
            y = torch.full((n, oh, ow, cout), float("nan"), dtype=torch.float16, device=dev)
            ep = L.Epilogue(1, 0.2, None, bias.data_ptr(), None, 0, None, None, None, None)
            L.check(lib.dsr_conv_fwd(C.byref(d), x.data_ptr(), wf.data_ptr(), C.byref(ep), y.data_ptr(), st))
            torch.cuda.synchronize()
            outs.append(y)
    finally:
        if old is None:
            os.environ.pop("DSR_CONV_PADX", None)
        else:
            os.environ["DSR_CONV_PADX"] = old
    assert torch.isfinite(outs[0].float()).all()
    assert torch.equal(outs[0], outs[1])
    # and against torch on the padded image
    xp = TF.pad(x.float().permute(0, 3, 1, 2).cpu(), (1, 1, 1, 1), mode="reflect" if pmode == 1 else "replicate")
    ref = TF.leaky_relu(TF.conv2d(xp, wt.half().float().cpu(), bias.cpu(), stride=stride), 0.2)
    got = outs[1].float().permute(0, 3, 1, 2).cpu()
    assert (got - ref).abs().max().item() <= 1.2e-2 * ref.abs().max().item()


@pytest.mark.parametrize("op", ["fwd", "fwd_stats", "dgrad", "dgrad_masked", "dgrad_s2"])
def test_conv_256x256_tile_equals_128x128(dev, op):
    """The big-tile variants of the gather kernel (8 waves, one block per CU; conv_gemm.hip) at a shape that dispatches to them
    (314 tiles of 256 rows): the 256x256 tile and the 224x256 tile (7 instead of 8 m-tiles per wave: taken by launches
    without BatchNorm statistics whose 256-row tiles would leave the last round of the chip mostly idle) walk K in the same
    order as the 128x128 tile (DSR_CONV_BIG=0): BIT FOR BIT.  And ALL of them against a plain fp32 PyTorch reference of the
    same op on the same bf16 operands (conv2d / conv_transpose2d on the CPU; BatchNorm sums of the REFERENCE's output), which
    is what ties the dominant kernel of the batch-32 step to something other than another kernel of this library."""
    import ctypes as C
    import os
    L = P("_lib")
    lib = L.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    n, h, w, cin, cout, stride = (3, 160, 167, 128, 256, 1) if op != "dgrad_s2" else (3, 320, 334, 256, 64, 2)
    if op in ("dgrad", "dgrad_masked"):
        cin, cout = 256, 128
    d = L.ConvDesc(L.BF16, n, h, w, cin, cout, 3, 3, stride, 1, 0)
    oh, ow = (h + 2 - 3) // stride + 1, (w + 2 - 3) // stride + 1
    g = torch.Generator(device="cpu").manual_seed(7)
    wt = ((torch.rand(cout, cin, 3, 3, generator=g) - 0.5) * 0.1).to(dev)
    wf = torch.empty(lib.dsr_conv_packed_elems(C.byref(d), 0), dtype=torch.bfloat16, device=dev)
    wd = torch.empty(lib.dsr_conv_packed_elems(C.byref(d), 1), dtype=torch.bfloat16, device=dev)
    L.check(lib.dsr_conv_pack_weight(C.byref(d), wt.data_ptr(), wf.data_ptr(), wd.data_ptr(), st))
    x = (torch.rand(n, h, w, cin, generator=g) - 0.5).to(torch.bfloat16).to(dev)
    dy = (torch.rand(n, oh, ow, cout, generator=g) - 0.5).to(torch.bfloat16).to(dev)
    bias = (torch.rand(cout, generator=g) - 0.5).to(dev)
    outs, stats = {}, {}
    keys = ("DSR_CONV_BIG", "DSR_DGRAD_S2", "DSR_CONV_BM224", "DSR_CONV_BM64")
    old = {k: os.environ.get(k) for k in keys}
    os.environ["DSR_DGRAD_S2"] = "0"       # this test is about the gather kernel's tiles: keep stride-2 dgrads on it
    try:
        # DSR_CONV_BM224: 0 = never, 1 = where it saves rounds (default), 2 = wherever the 256x256 tile would be taken
        # DSR_CONV_BM64: 0 = never, 1 = launches of fewer than 256 tiles (default), 2 = wherever the 128x128 tile would be taken
        for mode, big, b224, b64, want in (("t128", "0", "0", "0", "128x128"), ("t256", "2", "0", "0", "256x256"),
                                           ("t224", "2", "2", "0", "224x256"), ("t64", "0", "0", "2", "64x128")):
            if mode in ("t224", "t64") and op == "fwd_stats":
                continue                   # (the 224- and 64-row tiles carry no statistics epilogue)
            os.environ["DSR_CONV_BIG"], os.environ["DSR_CONV_BM224"], os.environ["DSR_CONV_BM64"] = big, b224, b64
            if op in ("fwd", "fwd_stats"):
                y = torch.full((n, oh, ow, cout), float("nan"), dtype=torch.bfloat16, device=dev)
                rows = lib.dsr_conv_stats_rows(C.byref(d))
                part = torch.full(((rows + 64) * 2 * cout,), float("nan"), dtype=torch.float32, device=dev)
                ep = L.Epilogue(L.ACT_RELU if op == "fwd" else L.ACT_NONE, 0.0, None, bias.data_ptr(),
                                part.data_ptr() if op == "fwd_stats" else None, 0, None)
                name = lib.dsr_conv_kernel_name(C.byref(d), 0, C.byref(ep)).decode()
                L.check(lib.dsr_conv_fwd(C.byref(d), x.data_ptr(), wf.data_ptr(), C.byref(ep), y.data_ptr(), st))
                outs[mode] = y
                if op == "fwd_stats":     # BatchNorm statistics: per-channel sum and sum of squares over all pixels
                    stats[mode] = part[:rows * 2 * cout].reshape(rows, 2, cout).double().sum(0)
            else:
                dx = torch.full((n, h, w, cin), float("nan"), dtype=torch.bfloat16, device=dev)
                name = lib.dsr_conv_kernel_name(C.byref(d), 1, None).decode()
                if op == "dgrad_masked":       # dx * relu'(x): the activation mask folded into the store loop
                    L.check(lib.dsr_conv_dgrad_masked(C.byref(d), dy.data_ptr(), wd.data_ptr(), x.data_ptr(), L.ACT_RELU, 0.0,
                                                      dx.data_ptr(), st))
                else:
                    wsz = lib.dsr_conv_dgrad_workspace(C.byref(d))
                    ws = torch.empty(max(wsz, 16), dtype=torch.uint8, device=dev)
                    L.check(lib.dsr_conv_dgrad(C.byref(d), dy.data_ptr(), wd.data_ptr(), dx.data_ptr(), ws.data_ptr(), wsz, st))
                outs[mode] = dx
            assert want in name, (mode, name)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    torch.cuda.synchronize()
    assert all(torch.isfinite(o.float()).all() for o in outs.values())
    assert torch.equal(outs["t128"], outs["t256"])
    for extra in ("t224", "t64"):
        if extra in outs:
            assert torch.equal(outs["t128"], outs[extra]), extra
    # ---- against a plain fp32 PyTorch reference of the same op on the same bf16 operands, at THIS shape: one bf16 rounding
    # of the output = 2^-9 relative to the value, stated relative to the tensor's maximum as everywhere in this file
    xr, wr, dyr = x.float().cpu().permute(0, 3, 1, 2), wt.to(torch.bfloat16).float().cpu(), dy.float().cpu().permute(0, 3, 1, 2)
    if op in ("fwd", "fwd_stats"):
        pre = TF.conv2d(xr, wr, bias.cpu(), stride=stride, padding=1)
        ref = torch.relu(pre) if op == "fwd" else pre
    else:
        ref = TF.conv_transpose2d(dyr, wr, stride=stride, padding=1, output_padding=(h + 2 - 3) % stride if stride > 1 else 0)
        if op == "dgrad_masked":
            ref = ref * (xr > 0).float()
    for mode, o in outs.items():
        got = o.float().cpu().permute(0, 3, 1, 2)
        assert got.shape == ref.shape
        assert rel_err(got, ref) <= 1.2e-2, mode
    if stats:    # same fp32 accumulators, summed in a different order (channel-major vs pixel-major epilogue)
        assert all(torch.isfinite(s_).all() for s_ in stats.values())
        assert float((stats["t128"] - stats["t256"]).abs().max() / stats["t128"].abs().max()) < 1e-5
        # per-channel sum / sum of squares of the REFERENCE's pre-activation output (fp32 conv on the CPU), not of the kernel's own
        ref_st = torch.stack([pre.double().sum((0, 2, 3)), (pre.double() ** 2).sum((0, 2, 3))])
        for mode, s_ in stats.items():
            assert float((s_.cpu() - ref_st).abs().max() / ref_st.abs().max()) < 1e-4, mode


@pytest.mark.parametrize("n,h,w,cout", [(3, 40, 100, 256), (2, 8, 32, 128), (1, 67, 130, 192), (40, 16, 64, 256)])
def test_conv_halo64_dgrad_vs_gather_kernel_and_fp32(dev, n, h, w, cout):
    """conv_halo64_kernel (input gradient of a 3x3 stride-1 layer with 64 inputs and `cout` = 128 / 192 / 256 outputs: a
    64-output convolution over `cout` channels, halo staged per 32-channel K-block; generator.py:30, discriminator.py:31)
    against the gather kernel it replaces (DSR_CONV_HALO64=0: the same products, summed tap-major there and channel-block-major
    here, so equal within fp32 summation order) and against a float64 conv_transpose2d of the same bf16 operands.  Shapes:
    ragged right / bottom tiles, the smallest admitted map, 192 = six K-blocks, and 640 tiles on 256 persistent blocks."""
    import ctypes as C
    import os
    L = P("_lib")
    lib = L.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    cin = 64
    d = L.ConvDesc(L.BF16, n, h, w, cin, cout, 3, 3, 1, 1, 0)
    g = torch.Generator(device="cpu").manual_seed(5)
    wt = bfr((torch.rand(cout, cin, 3, 3, generator=g) - 0.5) * 0.2)
    wf = torch.empty(lib.dsr_conv_packed_elems(C.byref(d), 0), dtype=torch.bfloat16, device=dev)
    wd = torch.empty(lib.dsr_conv_packed_elems(C.byref(d), 1), dtype=torch.bfloat16, device=dev)
    L.check(lib.dsr_conv_pack_weight(C.byref(d), wt.to(dev).data_ptr(), wf.data_ptr(), wd.data_ptr(), st))
    dy = bfr(torch.rand(n, cout, h, w, generator=g) - 0.5)
    dyg = dy.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(dev)
    outs = {}
    old = os.environ.get("DSR_CONV_HALO64")
    try:
        for mode in ("1", "0"):
            os.environ["DSR_CONV_HALO64"] = mode
            name = lib.dsr_conv_kernel_name(C.byref(d), 1, None).decode()
            assert ("halo64" in name) == (mode == "1"), (mode, name)
            dx = torch.full((n, h, w, cin), float("nan"), dtype=torch.bfloat16, device=dev)
            L.check(lib.dsr_conv_dgrad(C.byref(d), dyg.data_ptr(), wd.data_ptr(), dx.data_ptr(), None, 0, st))
            outs[mode] = dx
    finally:
        if old is None:
            os.environ.pop("DSR_CONV_HALO64", None)
        else:
            os.environ["DSR_CONV_HALO64"] = old
    torch.cuda.synchronize()
    assert torch.isfinite(outs["1"].float()).all()
    ref = TF.conv_transpose2d(dy.double(), wt.double(), padding=1)
    for mode, o in outs.items():
        assert rel_err(o.float().cpu().permute(0, 3, 1, 2), ref) <= 6e-3, mode       # one bf16 rounding of the output
    assert rel_err(outs["1"].float(), outs["0"].float()) <= 2.0 ** -7
    assert float((outs["1"].float() != outs["0"].float()).float().mean()) < 0.05          # (different summation order: a few last bits)


@pytest.mark.parametrize("n,h,w,cin,cout", [(2, 40, 100, 128, 128), (1, 67, 130, 256, 128), (3, 16, 64, 128, 256), (36, 16, 64, 128, 128)])
def test_conv_halo64_two_slices_of_64_outputs(dev, n, h, w, cin, cout):
    """conv_halo64_kernel on a layer with 128 outputs (forward: VGG conv2_2, utils/GAN.py:26, with bias + ReLU in the epilogue)
    or 128 inputs (input gradient, plain and with the ReLU mask of the activation in front folded into its stores): two
    64-channel slices per spatial tile (opt-in, DSR_CONV_HALO64=2: faster launch by launch, slower inside the two-stream step).
    Against the gather kernel (the default for these layers: the same products in another order) and float64 conv2d / conv_transpose2d on the same bf16 operands.  Ragged tiles, K = 128 and 256,
    more (tile, slice) pairs than persistent blocks."""
    import ctypes as C
    import os
    L = P("_lib")
    F = P("functional")
    lib = L.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    d = L.ConvDesc(L.BF16, n, h, w, cin, cout, 3, 3, 1, 1, 0)
    g = torch.Generator(device="cpu").manual_seed(cin + cout + h)
    wt = bfr((torch.rand(cout, cin, 3, 3, generator=g) - 0.5) * 0.1)
    bias = (torch.rand(cout, generator=g) - 0.5) * 0.2
    x = bfr(torch.rand(n, cin, h, w, generator=g) - 0.5)
    dy = bfr(torch.rand(n, cout, h, w, generator=g) - 0.5)
    wf = torch.empty(lib.dsr_conv_packed_elems(C.byref(d), 0), dtype=torch.bfloat16, device=dev)
    wd = torch.empty(lib.dsr_conv_packed_elems(C.byref(d), 1), dtype=torch.bfloat16, device=dev)
    L.check(lib.dsr_conv_pack_weight(C.byref(d), wt.to(dev).data_ptr(), wf.data_ptr(), wd.data_ptr(), st))
    xg, dyg, bg = to_nhwc(x, cin).to(dev), to_nhwc(dy, cout).to(dev), bias.to(dev)
    ep = L.Epilogue(L.ACT_RELU, 0.0, None, bg.data_ptr(), None, 0, None)
    do_fwd, do_dgrad = cout == 128, cin == 128
    outs = {}
    old = os.environ.get("DSR_CONV_HALO64")
    try:
        for mode in ("2", "1"):
            os.environ["DSR_CONV_HALO64"] = mode
            res = {}
            if do_fwd:
                name = lib.dsr_conv_kernel_name(C.byref(d), 0, C.byref(ep)).decode()
                assert ("halo64" in name) == (mode == "2"), (mode, name)
                y = torch.full((n, h, w, cout), float("nan"), dtype=torch.bfloat16, device=dev)
                L.check(lib.dsr_conv_fwd(C.byref(d), xg.data_ptr(), wf.data_ptr(), C.byref(ep), y.data_ptr(), st))
                res["y"] = y
            if do_dgrad:
                name = lib.dsr_conv_kernel_name(C.byref(d), 1, None).decode()
                assert ("halo64" in name) == (mode == "2"), (mode, name)
                dx = torch.full((n, h, w, cin), float("nan"), dtype=torch.bfloat16, device=dev)
                L.check(lib.dsr_conv_dgrad(C.byref(d), dyg.data_ptr(), wd.data_ptr(), dx.data_ptr(), None, 0, st))
                res["dx"] = dx
                if lib.dsr_conv_dgrad_masked_supported(C.byref(d)):
                    dxm = torch.full((n, h, w, cin), float("nan"), dtype=torch.bfloat16, device=dev)
                    L.check(lib.dsr_conv_dgrad_masked(C.byref(d), dyg.data_ptr(), wd.data_ptr(), xg.data_ptr(), L.ACT_RELU, 0.0,
                                                      dxm.data_ptr(), st))
                    res["dxm"] = dxm
            outs[mode] = res
    finally:
        if old is None:
            os.environ.pop("DSR_CONV_HALO64", None)
        else:
            os.environ["DSR_CONV_HALO64"] = old
    torch.cuda.synchronize()
    refs = {}
    if do_fwd:
        refs["y"] = TF.relu(TF.conv2d(x.double(), wt.double(), bias.double(), padding=1))
    if do_dgrad:
        refs["dx"] = TF.conv_transpose2d(dy.double(), wt.double(), padding=1)
        refs["dxm"] = refs["dx"] * (x.double() > 0)
    for key in outs["2"]:
        a_, b_ = outs["2"][key].float().cpu(), outs["1"][key].float().cpu()
        assert torch.isfinite(a_).all(), key
        ck = cin if key != "y" else cout
        for o in (a_, b_):
            assert rel_err(from_nhwc(o, ck), refs[key]) <= 6e-3, key                        # one bf16 rounding of the output
        assert rel_err(a_, b_) <= 2.0 ** -7 and float((a_ != b_).float().mean()) < 0.05, key  # (another summation order)


@pytest.mark.parametrize("n,h,w,cin,cout", [(2, 64, 96, 64, 64), (1, 50, 38, 128, 192), (3, 32, 32, 256, 64)])
def test_conv_dgrad_s2_single_launch_equals_four_launches(dev, n, h, w, cin, cout):
    """conv_dgrad_s2_kernel (3x3 stride 2 pad 1 input gradient, all four output-parity classes from one staged dY tile;
    discriminator.py:29-35) against the four gather-kernel launches it replaces: same products in the same order, so BIT FOR
    BIT equal, ragged M tail included; and against a float64 conv_transpose on the same bf16 operands."""
    import ctypes as C
    import os
    L = P("_lib")
    lib = L.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    d = L.ConvDesc(L.BF16, n, h, w, cin, cout, 3, 3, 2, 1, 0)
    oh, ow = h // 2, w // 2
    g = torch.Generator(device="cpu").manual_seed(11)
    wt = bfr((torch.rand(cout, cin, 3, 3, generator=g) - 0.5) * 0.2)
    wf = torch.empty(lib.dsr_conv_packed_elems(C.byref(d), 0), dtype=torch.bfloat16, device=dev)
    wd = torch.empty(lib.dsr_conv_packed_elems(C.byref(d), 1), dtype=torch.bfloat16, device=dev)
    wdev = wt.to(dev)
    L.check(lib.dsr_conv_pack_weight(C.byref(d), wdev.data_ptr(), wf.data_ptr(), wd.data_ptr(), st))
    dy = bfr(torch.rand(n, cout, oh, ow, generator=g) - 0.5)
    dyd = dy.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(dev)
    outs = []
    old = os.environ.get("DSR_DGRAD_S2")
    try:
        for mode in ("0", "2"):        # 0: four parity-class launches, 2: the single-launch kernel whatever the grid size
            os.environ["DSR_DGRAD_S2"] = mode
            dx = torch.full((n, h, w, cin), float("nan"), dtype=torch.bfloat16, device=dev)
            name = lib.dsr_conv_kernel_name(C.byref(d), 1, None).decode()
            assert (name == "conv_dgrad_s2_kernel") == (mode == "2"), (mode, name)
            L.check(lib.dsr_conv_dgrad(C.byref(d), dyd.data_ptr(), wd.data_ptr(), dx.data_ptr(), None, 0, st))
            outs.append(dx)
    finally:
        if old is None:
            os.environ.pop("DSR_DGRAD_S2", None)
        else:
            os.environ["DSR_DGRAD_S2"] = old
    torch.cuda.synchronize()
    assert torch.isfinite(outs[1].float()).all()
    assert torch.equal(outs[0], outs[1])
    ref = TF.conv_transpose2d(dy.double(), wt.double(), stride=2, padding=1, output_padding=1)
    got = outs[1].float().permute(0, 3, 1, 2).cpu().double()
    assert tuple(ref.shape) == tuple(got.shape)
    assert float((got - ref).abs().max() / ref.abs().max()) < 1.2e-2


def test_conv_wgrad_batched_equals_per_layer_launches(dev):
    """dsr_conv_wgrad_batched (every 3x3 stride-1 weight gradient of a backward pass in one grouped launch) against
    dsr_conv_wgrad per layer and a float64 reference: mixed shapes (one 64x64 tile pair up to 3x2 pairs, ragged image
    sizes, reflect padding as in models/DIP/skip.py), two entries that share one dw (a weight applied to two batches:
    discriminator on real + generated, train_GAN.py:44-47 -> the SUM of both gradients), and more entries than one launch
    holds (chunking).  fp32 sums in a different split order: equal to ~1e-6 relative, not bitwise."""
    import ctypes as C
    L = P("_lib")
    lib = L.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device="cpu").manual_seed(5)
    shapes = [(2, 24, 40, 64, 64, 0), (1, 17, 33, 64, 128, 0), (2, 16, 16, 192, 96, 0), (1, 20, 28, 32, 64, 1),
              (2, 24, 40, 64, 64, 0)] + [(1, 12, 20, 64, 64, 0)] * 40      # 45 weights > DSR_WGRAD_BATCH_MAX
    descs, xs, dys, dws, refs = [], [], [], [], []
    for i, (n, h, w, cin, cout, pm) in enumerate(shapes):
        uses = 2 if i == 1 else 1                        # weight 1 is applied to two batches
        dw = torch.full((cout, cin, 3, 3), float("nan"), dtype=torch.float32, device=dev)
        ref = torch.zeros(cout, cin, 3, 3, dtype=torch.float64)
        for u in range(uses):
            nn_ = n + u                                  # (the two batches need not have one size)
            x = bfr(torch.rand(nn_, cin, h, w, generator=g) - 0.5)
            dy = bfr(torch.rand(nn_, cout, h, w, generator=g) - 0.5)
            xp = TF.pad(x.double(), (1, 1, 1, 1), mode="reflect" if pm == 1 else "constant")
            ref += torch.nn.grad.conv2d_weight(xp, (cout, cin, 3, 3), dy.double())
            descs.append(L.ConvDesc(L.BF16, nn_, h, w, cin, cout, 3, 3, 1, 1, pm))
            xs.append(x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(dev))
            dys.append(dy.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(dev))
            dws.append(dw)
        refs.append((dw, ref))
    k = len(descs)
    assert all(lib.dsr_conv_wgrad_batchable(C.byref(d)) == 1 for d in descs)
    assert lib.dsr_conv_wgrad_batchable(C.byref(L.ConvDesc(L.BF16, 1, 16, 16, 64, 64, 3, 3, 2, 1, 0))) == 0
    darr = (L.ConvDesc * k)(*descs)
    xa = (C.c_void_p * k)(*[t.data_ptr() for t in xs])
    ya = (C.c_void_p * k)(*[t.data_ptr() for t in dys])
    wa = (C.c_void_p * k)(*[t.data_ptr() for t in dws])
    wsz = lib.dsr_conv_wgrad_batched_workspace(k, darr, wa)
    assert wsz > 0
    ws = torch.empty(wsz, dtype=torch.uint8, device=dev)
    L.check(lib.dsr_conv_wgrad_batched(k, darr, xa, ya, wa, ws.data_ptr(), wsz, st))
    assert lib.dsr_conv_wgrad_batched(k, darr, xa, ya, wa, ws.data_ptr(), wsz // 2, st) < 0      # workspace too small
    torch.cuda.synchronize()
    # per-layer launches of the same problems
    single = []
    for d, x, dy in zip(descs, xs, dys):
        o = torch.empty(d.Cout, d.Cin, 3, 3, dtype=torch.float32, device=dev)
        w1 = lib.dsr_conv_wgrad_workspace(C.byref(d))
        s1 = torch.empty(w1, dtype=torch.uint8, device=dev)
        L.check(lib.dsr_conv_wgrad(C.byref(d), x.data_ptr(), dy.data_ptr(), o.data_ptr(), s1.data_ptr(), w1, st))
        single.append(o)
    torch.cuda.synchronize()
    j = 0
    for i, (dw, ref) in enumerate(refs):
        uses = 2 if i == 1 else 1
        per_layer = sum(single[j:j + uses])
        j += uses
        got = dw.cpu().double()
        assert torch.isfinite(got).all(), i
        scale = float(ref.abs().max())
        assert float((got - ref).abs().max()) < 2e-5 * scale + 1e-4, i          # fp32 accumulation of exact bf16 products
        assert float((got - per_layer.cpu().double()).abs().max()) < 2e-5 * scale + 1e-4, i


@pytest.mark.parametrize("n,h,w,cin,cout", [(2, 40, 48, 64, 128), (1, 33, 70, 128, 256), (3, 16, 16, 192, 384)])
def test_conv_wgrad_stride2_128_channel_blocks(dev, n, h, w, cin, cout):
    """conv_wgrad_dma_s2_kernel<COH = 2> (3x3 stride-2 weight gradient with 128 output channels per 8-wave block, so that the
    input halo is fetched once per 128 instead of once per 64 output channels; discriminator.py:31 at config 3) against the
    64-channel form (DSR_WGRAD_S2_CO128=0: the same per-wave products, a different pixel partition) and a float64
    torch.nn.grad.conv2d_weight of the same bf16 operands.  Odd sizes, a ragged last column tile, 192 input channels."""
    import ctypes as C
    import os
    L = P("_lib")
    lib = L.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    d = L.ConvDesc(L.BF16, n, h, w, cin, cout, 3, 3, 2, 1, 0)
    oh, ow = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    g = torch.Generator(device="cpu").manual_seed(h * w + cin)
    x = bfr(torch.rand(n, cin, h, w, generator=g) - 0.5)
    dy = bfr(torch.rand(n, cout, oh, ow, generator=g) - 0.5)
    ref = torch.nn.grad.conv2d_weight(x.double(), (cout, cin, 3, 3), dy.double(), stride=2, padding=1)
    xg = x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(dev)
    dyg = dy.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(dev)
    outs = {}
    old = os.environ.get("DSR_WGRAD_S2_CO128")
    try:
        for mode in ("2", "0"):
            os.environ["DSR_WGRAD_S2_CO128"] = mode
            dw = torch.full((cout, cin, 3, 3), float("nan"), dtype=torch.float32, device=dev)
            wsz = lib.dsr_conv_wgrad_workspace(C.byref(d))
            ws = torch.empty(wsz, dtype=torch.uint8, device=dev)
            L.check(lib.dsr_conv_wgrad(C.byref(d), xg.data_ptr(), dyg.data_ptr(), dw.data_ptr(), ws.data_ptr(), wsz, st))
            outs[mode] = dw.cpu().double()
    finally:
        if old is None:
            os.environ.pop("DSR_WGRAD_S2_CO128", None)
        else:
            os.environ["DSR_WGRAD_S2_CO128"] = old
    scale = float(ref.abs().max())
    for mode, got in outs.items():
        assert torch.isfinite(got).all(), mode
        assert float((got - ref).abs().max()) < 2e-5 * scale + 1e-4, mode         # fp32 accumulation of exact bf16 products
    assert float((outs["2"] - outs["0"]).abs().max()) < 2e-5 * scale + 1e-4


def test_conv_wgrad_batched_trunk_sized_group_vs_float64(dev):
    """The grouped weight-gradient launch at the problem COUNT and map sizes of a real backward pass (the generator's
    backward groups 35 problems; here 34: 30 trunk-shaped 64 -> 64 layers on 64x64 maps, a 64 -> 256 PixelShuffle-conv
    shape on 96x80, a 64 -> 128 and a 128 -> 256 layer, and one weight applied to two batches) against a float64
    torch.nn.grad.conv2d_weight of the same bf16 operands: fp32 accumulation of exact products, 2e-5 of the tensor's maximum."""
    import ctypes as C
    L = P("_lib")
    lib = L.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device="cpu").manual_seed(23)
    shapes = [(2, 64, 64, 64, 64)] * 30 + [(1, 96, 80, 64, 256), (2, 72, 64, 64, 128), (1, 64, 64, 128, 256)]
    descs, xs, dys, dws, refs = [], [], [], [], []
    for i, (n, h, w, cin, cout) in enumerate(shapes):
        uses = 2 if i == 3 else 1
        dw = torch.full((cout, cin, 3, 3), float("nan"), dtype=torch.float32, device=dev)
        ref = torch.zeros(cout, cin, 3, 3, dtype=torch.float64)
        for u in range(uses):
            x = bfr(torch.rand(n + u, cin, h, w, generator=g) - 0.5)
            dy = bfr(torch.rand(n + u, cout, h, w, generator=g) - 0.5)
            ref += torch.nn.grad.conv2d_weight(x.double(), (cout, cin, 3, 3), dy.double(), padding=1)
            descs.append(L.ConvDesc(L.BF16, n + u, h, w, cin, cout, 3, 3, 1, 1, 0))
            xs.append(x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(dev))
            dys.append(dy.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(dev))
            dws.append(dw)
        refs.append((dw, ref))
    k = len(descs)
    assert k == 34 and k > 30
    darr = (L.ConvDesc * k)(*descs)
    xa = (C.c_void_p * k)(*[t.data_ptr() for t in xs])
    ya = (C.c_void_p * k)(*[t.data_ptr() for t in dys])
    wa = (C.c_void_p * k)(*[t.data_ptr() for t in dws])
    wsz = lib.dsr_conv_wgrad_batched_workspace(k, darr, wa)
    assert wsz > 0
    ws = torch.empty(wsz, dtype=torch.uint8, device=dev)
    L.check(lib.dsr_conv_wgrad_batched(k, darr, xa, ya, wa, ws.data_ptr(), wsz, st))
    torch.cuda.synchronize()
    for i, (dw, ref) in enumerate(refs):
        got = dw.cpu().double()
        assert torch.isfinite(got).all(), i
        assert float((got - ref).abs().max()) < 2e-5 * float(ref.abs().max()) + 1e-4, i


def test_conv_dgrad_add_equals_dgrad_then_add(dev):
    """dsr_conv_dgrad_add (64 -> 64 3x3 input gradient with the skip path's gradient added in the epilogue; the residual blocks
    of generator.py:4-25) against dsr_conv_dgrad followed by a bf16 add: both round the conv sum to bf16, add in fp32 and round
    again => BIT FOR BIT equal, ragged tiles included; other shapes are refused."""
    import ctypes as C
    L = P("_lib")
    lib = L.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    n, h, w = 3, 37, 70
    d = L.ConvDesc(L.BF16, n, h, w, 64, 64, 3, 3, 1, 1, 0)
    assert lib.dsr_conv_dgrad_add_supported(C.byref(d)) == 1
    assert lib.dsr_conv_dgrad_add_supported(C.byref(L.ConvDesc(L.BF16, n, h, w, 64, 128, 3, 3, 1, 1, 0))) == 0
    g = torch.Generator(device="cpu").manual_seed(23)
    wt = ((torch.rand(64, 64, 3, 3, generator=g) - 0.5) * 0.2).to(dev)
    wf = torch.empty(lib.dsr_conv_packed_elems(C.byref(d), 0), dtype=torch.bfloat16, device=dev)
    wd = torch.empty(lib.dsr_conv_packed_elems(C.byref(d), 1), dtype=torch.bfloat16, device=dev)
    L.check(lib.dsr_conv_pack_weight(C.byref(d), wt.data_ptr(), wf.data_ptr(), wd.data_ptr(), st))
    dy = (torch.rand(n, h, w, 64, generator=g) - 0.5).to(torch.bfloat16).to(dev)
    add = (torch.rand(n, h, w, 64, generator=g) - 0.5).to(torch.bfloat16).to(dev)
    dx0 = torch.empty_like(dy)
    L.check(lib.dsr_conv_dgrad(C.byref(d), dy.data_ptr(), wd.data_ptr(), dx0.data_ptr(), None, 0, st))
    ref = dx0 + add
    dx1 = torch.full_like(dy, float("nan"))
    L.check(lib.dsr_conv_dgrad_add(C.byref(d), dy.data_ptr(), wd.data_ptr(), add.data_ptr(), dx1.data_ptr(), st))
    torch.cuda.synchronize()
    assert torch.isfinite(dx1.float()).all() and torch.equal(ref, dx1)
    bad = L.ConvDesc(L.BF16, n, h, w, 64, 128, 3, 3, 1, 1, 0)
    assert lib.dsr_conv_dgrad_add(C.byref(bad), dy.data_ptr(), wd.data_ptr(), add.data_ptr(), dx1.data_ptr(), st) < 0


@pytest.mark.parametrize("n,h,w,cin,cout", [(2, 37, 70, 64, 64), (2, 28, 28, 128, 128), (3, 20, 24, 256, 512), (2, 30, 42, 64, 128),
                                          (1, 14, 14, 512, 512)])
def test_conv_dgrad_masked_equals_dgrad_then_act_bwd(dev, n, h, w, cin, cout):
    """dsr_conv_dgrad_masked (the ReLU / LeakyReLU backward of the layer in front folded into the input-gradient store loop;
    the conv + ReLU chain of utils/GAN.py:19-57) against dsr_conv_dgrad followed by dsr_pw_act_bwd: the same fp32 product on
    the same rounded values => BIT FOR BIT, on the 64->64 kernel (residual-prefetch mode) and the 128x64 / 128x128 / 256x256
    tiles of the gather kernel, ragged sizes included."""
    import ctypes as C
    L = P("_lib")
    lib = L.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    d = L.ConvDesc(L.BF16, n, h, w, cin, cout, 3, 3, 1, 1, 0)
    assert lib.dsr_conv_dgrad_masked_supported(C.byref(d)) == 1
    g = torch.Generator(device="cpu").manual_seed(31)
    wt = ((torch.rand(cout, cin, 3, 3, generator=g) - 0.5) * 0.1).to(dev)
    wf = torch.empty(lib.dsr_conv_packed_elems(C.byref(d), 0), dtype=torch.bfloat16, device=dev)
    wd = torch.empty(lib.dsr_conv_packed_elems(C.byref(d), 1), dtype=torch.bfloat16, device=dev)
    L.check(lib.dsr_conv_pack_weight(C.byref(d), wt.data_ptr(), wf.data_ptr(), wd.data_ptr(), st))
    dy = (torch.rand(n, h, w, cout, generator=g) - 0.5).to(torch.bfloat16).to(dev)
    x = torch.relu(torch.rand(n, h, w, cin, generator=g) - 0.4).to(torch.bfloat16).to(dev)      # a ReLU output: ~40 % zeros
    for act, slope in ((L.ACT_RELU, 0.0), (L.ACT_LEAKY, 0.2)):
        xa = x if act == L.ACT_RELU else (x - 0.05 * (x == 0)).to(torch.bfloat16)              # Leaky outputs: negatives where it was zero
        dx0 = torch.empty_like(x)
        L.check(lib.dsr_conv_dgrad(C.byref(d), dy.data_ptr(), wd.data_ptr(), dx0.data_ptr(), None, 0, st))
        ref = torch.empty_like(x)
        L.check(lib.dsr_pw_act_bwd(L.BF16, dx0.data_ptr(), xa.data_ptr(), ref.data_ptr(), n, h, w, cin, cin, 0, act, slope, None, 1,
                                   n * h * w, None, st))
        got = torch.full_like(x, float("nan"))
        L.check(lib.dsr_conv_dgrad_masked(C.byref(d), dy.data_ptr(), wd.data_ptr(), xa.data_ptr(), act, slope, got.data_ptr(), st))
        torch.cuda.synchronize()
        assert torch.isfinite(got.float()).all() and torch.equal(ref, got), (act, slope)
    assert lib.dsr_conv_dgrad_masked(C.byref(d), dy.data_ptr(), wd.data_ptr(), x.data_ptr(), L.ACT_LEAKY, -0.1, got.data_ptr(), st) < 0
    s2 = L.ConvDesc(L.BF16, n, h - h % 2, w - w % 2, cin, cout, 3, 3, 2, 1, 0)
    assert lib.dsr_conv_dgrad_masked_supported(C.byref(s2)) == 0


def test_maxpool_relu_bwd_equals_pool_bwd_then_mask(dev):
    """dsr_maxpool2_relu_bwd (conv + ReLU + MaxPool of the VGG trunk) == dsr_maxpool2_bwd followed by the ReLU mask, bit for bit,
    including windows whose four inputs are all zero (gradient dropped) and ties."""
    import ctypes as C
    L = P("_lib")
    lib = L.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device="cpu").manual_seed(33)
    n, h, w, c = 2, 18, 26, 64
    x = torch.relu(torch.rand(n, h, w, c, generator=g) - 0.7).to(torch.bfloat16).to(dev)       # 70 % zeros: many all-zero windows
    dy = (torch.rand(n, h // 2, w // 2, c, generator=g) - 0.5).to(torch.bfloat16).to(dev)
    a, b = torch.empty_like(x), torch.full_like(x, float("nan"))
    L.check(lib.dsr_maxpool2_bwd(L.BF16, x.data_ptr(), dy.data_ptr(), a.data_ptr(), n, h, w, c, st))
    L.check(lib.dsr_maxpool2_relu_bwd(L.BF16, x.data_ptr(), dy.data_ptr(), b.data_ptr(), n, h, w, c, st))
    torch.cuda.synchronize()
    assert torch.equal(a * (x > 0), b)
    assert float((a != b).float().mean()) > 0.01        # (the mask does something on this input)
