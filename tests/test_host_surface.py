"""CPU: the host-side mirror of the reference's surface -- install_dropin(), utils/DIP helpers against the golden
fixture and (when /root/reference is present, i.e. in the build container) against the reference's own functions,
reference-format checkpoints through evaluate.save_model / load_model and the reference's own save_model / load_model.
No device work: module construction, state_dicts and host glue only."""
import importlib
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = "deep-super-resolution_amd"
REF = "/root/reference"
needs_ref = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree exists only in the build container")


def P(sub):
    return importlib.import_module(PKG + "." + sub)


def run_py(code, with_ref):
    """Run a snippet in a fresh interpreter (install_dropin edits sys.modules; keep that out of the pytest process)."""
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    pre = f"import sys, importlib\nsys.path.insert(0, {ROOT!r})\n" + (f"sys.path.insert(0, {REF!r})\n" if with_ref else "")
    r = subprocess.run([sys.executable, "-c", pre + textwrap.dedent(code)], capture_output=True, text=True, env=env,
                       cwd="/tmp", timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    return r.stdout


@needs_ref
def test_install_dropin_keeps_reference_packages():
    """with the reference on sys.path, the lines of train_GAN.py:11-15 / DIP.py:11-15 / eval_GAN.py:11-14
    must all resolve: mirrored leaves to this package (since round 2 that includes utils.degradation and the top-level
    dataset module, whose own version needs torchvision), everything else (utils.common) to the reference's own modules."""
    out = run_py("""
        pk = importlib.import_module("deep-super-resolution_amd")
        names = pk.install_dropin()
        from models.GAN.discriminator import Discriminator
        from models.GAN.generator import Generator
        from utils.GAN import *
        from utils.common import *
        from utils.downsampler import Downsampler
        from models.DIP import get_net
        from utils.DIP import *
        import utils.common, utils.degradation, utils.DIP, models.DIP.skip, models.GAN.generator as gg
        assert utils.common.__file__.startswith("/root/reference/"), utils.common.__file__
        assert utils.degradation.__name__ == "deep-super-resolution_amd.utils.degradation"
        from dataset import GANDIV2KDataset, DIV2KDataset, get_image_pair          # (train_GAN.py:12 `from dataset import *`)
        assert GANDIV2KDataset.__module__ == "deep-super-resolution_amd.dataset"
        assert callable(utils.degradation.downsample) and utils.degradation.save_model is utils.common.save_model
        for obj in (Generator, Discriminator, Downsampler, get_net, PerceptualLoss, get_loss_D, optimize, get_noise):
            assert obj.__module__.startswith("deep-super-resolution_amd."), (obj, obj.__module__)
        assert gg.__name__ == "deep-super-resolution_amd.models.GAN.generator"
        assert models.DIP.skip.__name__ == "deep-super-resolution_amd.models.DIP.skip"
        # utils/DIP.py:3 re-exports utils.common: `from utils.DIP import *` users find save_model & co there
        assert utils.DIP.save_model is utils.common.save_model and callable(save_log) and callable(np_to_torch)
        g = Generator(factor=8, residual_blocks_count=1)
        assert type(g).__module__.startswith("deep-super-resolution_amd.")
        print("OK", len(names))
    """, with_ref=True)
    assert out.startswith("OK 10")


def test_install_dropin_standalone():
    """Without the reference on sys.path the aliased leaves still import (empty stand-in parent packages)."""
    out = run_py("""
        pk = importlib.import_module("deep-super-resolution_amd")
        pk.install_dropin()
        from models.GAN.generator import Generator
        from models.GAN.discriminator import Discriminator
        from models.DIP import get_net
        from utils.downsampler import Downsampler
        from utils.GAN import PerceptualLoss, get_adversarial_loss
        from utils.DIP import optimize, get_params, get_noise, fill_noise
        assert Generator.__module__.startswith("deep-super-resolution_amd.")
        try:
            import utils.common
        except ImportError:
            print("OK")
    """, with_ref=False)
    assert out.strip() == "OK"


# ----------------------------------------------------------------------------- utils/DIP.py helpers (row a13, f4)
def test_product_get_noise_matches_reference_fixture(golden):
    """The PRODUCT's get_noise (not the oracle's) against the tensor the reference's get_noise drew under seed 0
    (tests/golden/make_golden.py gen_noise): same CPU generator, same call order, bit for bit."""
    D = P("utils.DIP")
    torch.manual_seed(0)
    t = D.get_noise(32, "noise", (8, 12))
    assert t.dtype == torch.float32 and tuple(t.shape) == (1, 32, 8, 12)
    np.testing.assert_array_equal(t.numpy(), golden("get_noise_seed0")["t"])
    torch.manual_seed(0)
    sq = D.get_noise(32, "noise", 8)                       # int spatial size (utils/DIP.py:89-90)
    assert tuple(sq.shape) == (1, 32, 8, 8)
    torch.manual_seed(3)
    n1 = D.get_noise(4, "noise", (5, 7), noise_type="n", var=0.5)
    torch.manual_seed(3)
    assert torch.equal(n1, torch.zeros(1, 4, 5, 7).normal_() * 0.5)
    with pytest.raises(AssertionError):
        D.get_noise(3, "meshgrid", (4, 4))                 # input_depth must be 2
    with pytest.raises(AssertionError):
        D.get_noise(2, "nope", (4, 4))
    with pytest.raises(AssertionError):
        D.fill_noise(torch.zeros(2), "x")


@needs_ref
def test_dip_helpers_equal_reference_functions():
    """get_noise (both methods), fill_noise and get_params against the reference's own functions, imported here."""
    out = run_py("""
        import torch, numpy as np
        import utils.DIP as ref                                     # the reference's
        ours = importlib.import_module("deep-super-resolution_amd.utils.DIP")
        for args, kw in (((32, "noise", (8, 12)), {}), ((3, "noise", 16), dict(noise_type="n", var=0.25)),
                         ((2, "meshgrid", (6, 9)), {})):
            torch.manual_seed(5); a = ref.get_noise(*args, **kw)
            torch.manual_seed(5); b = ours.get_noise(*args, **kw)
            assert a.dtype == b.dtype and a.shape == b.shape and torch.equal(a, b), args
        net = torch.nn.Sequential(torch.nn.Conv2d(2, 3, 1), torch.nn.BatchNorm2d(3))
        down = torch.nn.Conv2d(3, 3, 2)
        for spec in ("net", "net,input", "input", "net,down", "down,net"):
            zi_a, zi_b = torch.zeros(1, 2, 4, 4), torch.zeros(1, 2, 4, 4)
            pa, pb = ref.get_params(spec, net, zi_a, down), ours.get_params(spec, net, zi_b, down)
            assert len(pa) == len(pb) and all(x is y or (x is zi_a and y is zi_b) for x, y in zip(pa, pb)), spec
            assert zi_a.requires_grad == zi_b.requires_grad
        for f in (ref.get_params, ours.get_params):
            try:
                f("bogus", net, None)
                raise SystemExit("no assert")
            except AssertionError:
                pass
        print("OK")
    """, with_ref=True)
    assert out.strip() == "OK"


def test_optimize_rejects_unknown_optimizer():
    D = P("utils.DIP")
    with pytest.raises(AssertionError):
        D.optimize("sgd", [torch.zeros(1, requires_grad=True)], lambda: None, 0.1, 1)


# ----------------------------------------------------------------------------- checkpoints (row f2)
def _seeded(cls, *a, **kw):
    torch.manual_seed(1234)
    return cls(*a, **kw)


@needs_ref
def test_checkpoint_roundtrip_with_reference_classes(tmp_path):
    """utils/common.py:11-18,46-60 both ways, with the reference's OWN classes and functions imported here:
    (1) reference Generator/Discriminator/DIP net -> reference save_model -> this package's load_model into the mirror;
    (2) the same file re-keyed with the DataParallel ``module.`` prefix;  (3) mirror -> evaluate.save_model -> reference
    load_model into the reference class (strict load_state_dict).  Keys, order, shapes, dtypes and values identical."""
    out = run_py(f"""
        import torch, os
        from collections import OrderedDict
        from models.GAN.generator import Generator as RefG
        from models.GAN.discriminator import Discriminator as RefD
        from models.DIP import get_net as ref_get_net
        import utils.common as refc
        ev = importlib.import_module("deep-super-resolution_amd.evaluate")
        G = importlib.import_module("deep-super-resolution_amd.models.GAN.generator").Generator
        D = importlib.import_module("deep-super-resolution_amd.models.GAN.discriminator").Discriminator
        get_net = importlib.import_module("deep-super-resolution_amd.models.DIP").get_net
        tmp = {str(tmp_path)!r}
        cases = [("g", lambda: RefG(8, 2), lambda: G(8, 2)), ("d", lambda: RefD((48, 32)), lambda: D((48, 32))),
                 ("dip", lambda: ref_get_net(8, "skip", "reflection", "bilinear", skip_n33d=16, skip_n33u=16, skip_n11=4, num_scales=3),
                  lambda: get_net(8, "skip", "reflection", "bilinear", skip_n33d=16, skip_n33u=16, skip_n11=4, num_scales=3))]
        for tag, mk_ref, mk_ours in cases:
            torch.manual_seed(7)
            ref = mk_ref()
            for b in ref.buffers():                       # non-trivial running statistics / counters
                if b.dtype.is_floating_point: b.uniform_(0.5, 1.5)
                else: b.fill_(3)
            refc.save_model(ref, tag, tmp)                # the reference's own writer
            path = os.path.join(tmp, tag + ".pth")
            rsd = ref.state_dict()
            ours = ev.load_model(mk_ours(), path)
            osd = ours.state_dict()
            assert list(osd.keys()) == list(rsd.keys()), tag
            for k in rsd:
                assert osd[k].dtype == rsd[k].dtype and osd[k].shape == rsd[k].shape and torch.equal(osd[k], rsd[k]), (tag, k)
            # a checkpoint of a DataParallel/DDP-wrapped model: every key carries "module."
            pref = os.path.join(tmp, tag + "_module.pth")
            torch.save(OrderedDict(("module." + k, v) for k, v in rsd.items()), pref)
            ours2 = ev.load_model(mk_ours(), pref)
            assert all(torch.equal(ours2.state_dict()[k], rsd[k]) for k in rsd), tag
            ref_via_ref = refc.load_model(mk_ref(), pref)  # the reference's reader agrees on that file
            assert all(torch.equal(ref_via_ref.state_dict()[k], rsd[k]) for k in rsd), tag
            # the other direction
            back = ev.save_model(ours, tag + "_back", tmp)
            ref2 = refc.load_model(mk_ref(), back)
            assert all(torch.equal(ref2.state_dict()[k], rsd[k]) for k in rsd), tag
        print("OK")
    """, with_ref=True)
    assert out.strip().endswith("OK")


def test_checkpoint_roundtrip_mirror_only(tmp_path):
    """evaluate.save_model / load_model on the mirror alone (runs anywhere): plain and ``module.``-prefixed files."""
    from collections import OrderedDict
    ev, Gm = P("evaluate"), P("models.GAN.generator")
    g = _seeded(Gm.Generator, 4, 1)
    path = ev.save_model(g, "g", str(tmp_path))
    assert path.endswith("g.pth")
    g2 = ev.load_model(Gm.Generator(4, 1), path)
    assert all(torch.equal(a, b) for a, b in zip(g.state_dict().values(), g2.state_dict().values()))
    pref = os.path.join(str(tmp_path), "g_module.pth")
    torch.save(OrderedDict(("module." + k, v) for k, v in g.state_dict().items()), pref)
    g3 = ev.load_model(Gm.Generator(4, 1), pref)
    assert list(g3.state_dict().keys()) == list(g.state_dict().keys())
    assert all(torch.equal(a, b) for a, b in zip(g.state_dict().values(), g3.state_dict().values()))
    with pytest.raises(RuntimeError):                      # wrong architecture: load_state_dict raises like the reference's
        ev.load_model(Gm.Generator(4, 2), path)


# ----------------------------------------------------------------------------- perceptual loss construction (row f3)
def test_perceptual_loss_accepts_local_vgg_state_dict():
    """utils/GAN.py:64-78 with a caller-supplied torchvision ``vgg19().features`` state_dict (keys "<i>.weight/bias"):
    key mapping net.0.<i>.*, every parameter frozen, values taken over; a wrong dict is rejected."""
    G = P("utils.GAN")
    gen = torch.Generator().manual_seed(5)
    idx = [0, 2, 5, 7, 10, 12, 14, 16, 19, 21, 23, 25, 28, 30, 32, 34]
    chans = [(3, 64), (64, 64), (64, 128), (128, 128), (128, 256), (256, 256), (256, 256), (256, 256), (256, 512),
             (512, 512), (512, 512), (512, 512), (512, 512), (512, 512), (512, 512), (512, 512)]
    feats = {}
    for i, (ci, co) in zip(idx, chans):
        feats[f"{i}.weight"] = torch.randn(co, ci, 3, 3, generator=gen) * 0.01
        feats[f"{i}.bias"] = torch.randn(co, generator=gen) * 0.01
    p = G.PerceptualLoss(vgg_state_dict=feats)
    sd = p.state_dict()
    assert list(sd.keys()) == [f"vgg_loss.net.0.{k}" for k in feats]          # utils/GAN.py:71-72 nesting
    assert all(torch.equal(sd[f"vgg_loss.net.0.{k}"], v) for k, v in feats.items())
    assert p.vgg_loss.pretrained and all(not q.requires_grad for q in p.parameters())   # :77-78
    bad = dict(feats)
    bad.pop("34.bias")
    with pytest.raises(RuntimeError):
        G.PerceptualLoss(vgg_state_dict=bad)
    assert not G.PerceptualLoss().vgg_loss.pretrained


def test_data_host_logic_matches_oracle():
    """Host-side pieces of the data path (no GPU): the product's Pillow coefficient tables (utils/degradation.resample_tables,
    what the HIP resampler is fed) against the oracle's restatement of Pillow's precompute_coeffs / normalize_coeffs_8bpc, and
    the patch sampler against dataset.py:121-147 as restated in the oracle."""
    import numpy as np
    from oracle import data as od
    D, DS = P("utils.degradation"), P("dataset")
    for n_in, n_out in [(124, 62), (90, 45), (53, 37), (31, 124), (2040, 1020), (339, 255)]:
        ksize, bounds, kk = D.resample_tables(n_in, n_out, "cpu")
        rk, rb, rkk = od.resample_coeffs(n_in, n_out)
        assert ksize == rk and np.array_equal(bounds.numpy(), rb) and np.array_equal(kk.numpy(), rkk), (n_in, n_out)
        assert int(kk.sum(1).min()) > (1 << 22) - 64 and int(kk.sum(1).max()) < (1 << 22) + 64      # weights sum to ~1.0 in 22-bit fixed point
    a, b = np.random.RandomState(9), np.random.RandomState(9)
    for _ in range(50):
        assert DS.train_patch_coords(170, 255, (128, 96), 4, a) == od.train_patch_coords(170, 255, 128, 96, 4, b)


def test_act_link_and_batched_wgrad_contexts_without_gpu():
    """functional.ActLink is a plain hand-over cell; functional.batched_wgrad is re-entrant-safe bookkeeping: an empty block
    launches nothing, a nested block defers to the outer one, an exception discards the batch."""
    F = P("functional")
    link = F.ActLink(F.ACT_RELU)
    assert (link.act, link.slope, link.premasked) == (F.ACT_RELU, 0.0, False)
    with F.batched_wgrad() as outer:
        assert F._wgrad_batch is outer
        with F.batched_wgrad() as inner:
            assert F._wgrad_batch is outer and inner is not outer
        assert F._wgrad_batch is outer
    assert F._wgrad_batch is None
    with F.batched_wgrad(False) as off:
        assert F._wgrad_batch is None and off.items == {}
    with pytest.raises(KeyError):
        with F.batched_wgrad():
            raise KeyError("x")
    assert F._wgrad_batch is None
    # a weight with a batchable and an unbatchable use (bookkeeping only; the launches are GPU tests): the unbatchable use
    # after a batched one is parked as an addend, the other order keeps the weight out of the batch altogether
    w1, w2 = torch.zeros(4, 4, 3, 3), torch.zeros(4, 4, 3, 3)
    xx = torch.zeros(1, 4, 4, 8)
    with pytest.raises(KeyError):
        with F.batched_wgrad() as b:
            ph = b.add(w1, None, xx, xx, (4, 4, 3, 3))
            assert ph is not None and b.add(w1, None, xx, xx, (4, 4, 3, 3)) is None
            extra = torch.ones(4, 4, 3, 3)
            assert b.add_unbatchable(w1, extra) is True and b.items[id(w1)][4] == [extra]
            assert b.add_unbatchable(w2, extra) is False and id(w2) in b.unbatched
            w1.grad = ph                                  # what autograd does with the placeholder
            raise KeyError("backward failed")             # (exit without a launch: needs no GPU)
    assert w1.grad is None and F._wgrad_batch is None
