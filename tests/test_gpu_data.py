"""SURVEY.md 8f row 1 on the GPU: the data-side byte kernels (csrc/data.hip) behind the reference's dataset.py /
utils/degradation.py surface, against the oracle (oracle/data.py, pinned to Pillow and to the reference's own
utils/degradation.py), the golden fixtures, and Pillow itself -- all uint8 / exact-float work, so every check is BIT FOR BIT."""
import importlib
import os

import numpy as np
import pytest
import torch

from oracle import data as od

pytestmark = pytest.mark.gpu
PKG = "deep-super-resolution_amd"
HERE = os.path.dirname(os.path.abspath(__file__))


def P(sub):
    return importlib.import_module(PKG + "." + sub)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def golden(name):
    return np.load(os.path.join(HERE, "golden", name + ".npz"))


def test_resize_matches_oracle_golden_and_pillow(dev):
    """degradation.downsample / resize (utils/degradation.py:19-20, dataset.py:21-45): tensors, arrays and PIL images in,
    the same type out; equal to the reference's own outputs (golden), to the oracle and to the Pillow installed here."""
    D = P("utils.degradation")
    z = golden("data_degradation")
    img = od.sample_image("in:data_a", 90, 124)
    t = torch.from_numpy(img).to(dev)
    d2 = D.downsample(t, 2)
    assert d2.dtype == torch.uint8 and d2.is_cuda and tuple(d2.shape) == (45, 62, 3)
    assert np.array_equal(d2.cpu().numpy(), z["down2"])
    d4 = D.downsample(d2)
    assert np.array_equal(d4.cpu().numpy(), z["down4"])
    assert np.array_equal(D.downsample(img, factor=3), z["down3"])                       # numpy in, numpy out
    hr = D.resize(t, 4 * d4.shape[1], 4 * d4.shape[0])
    assert np.array_equal(hr.cpu().numpy(), z["hr_resized"])
    assert np.array_equal(D.resize(od.sample_image("in:data_b", 71, 53), 37, 50), z["odd_resized"])
    from PIL import Image
    pil = Image.fromarray(img)
    out = D.downsample(pil, 2, Image.BICUBIC)                                             # PIL in, PIL out (the reference's call)
    assert isinstance(out, Image.Image) and np.array_equal(np.array(out), np.array(pil.resize((62, 45), Image.BICUBIC)))
    with pytest.raises(NotImplementedError):
        D.downsample(pil, 2, Image.BILINEAR)
    rng = np.random.RandomState(3)
    for h, w, ow, oh in [(64, 96, 48, 32), (37, 53, 18, 26), (100, 77, 77, 50), (50, 60, 100, 120), (45, 80, 31, 80),
                         (678, 1020, 339, 510)]:                                          # (the last: a DIV2K image halved)
        a = rng.randint(0, 256, (h, w, 3), dtype=np.uint8)
        got = D.resize(torch.from_numpy(a).to(dev), ow, oh).cpu().numpy()
        assert np.array_equal(got, np.array(Image.fromarray(a).resize((ow, oh), Image.BICUBIC))), (h, w, ow, oh)
        if h * w < 10000:
            assert np.array_equal(got, od.resize_u8(a, ow, oh))


def test_noise_matches_reference_draws(dev):
    """add_gaussian_noise / add_salt_pepper_noise with rng="numpy": numpy's global generator, the reference's draw order =>
    the reference's own pixels for the same seed (golden); rng="device" only has to respect the definition."""
    D = P("utils.degradation")
    z = golden("data_degradation")
    d2 = torch.from_numpy(z["down2"]).to(dev)
    np.random.seed(7)
    g = D.add_gaussian_noise(d2, std=0.1)
    assert np.array_equal(g.cpu().numpy(), z["gauss"])
    np.random.seed(8)
    sp = D.add_salt_pepper_noise(d2, s=0.02, p=0.03)
    assert np.array_equal(sp.cpu().numpy(), z["salt_pepper"])
    assert np.array_equal(d2.cpu().numpy(), z["down2"])                                  # inputs are not written
    np.random.seed(7)
    assert np.array_equal(D.add_gaussian_noise(z["down2"], std=0.1), z["gauss"])         # numpy in, numpy out
    torch.manual_seed(0)
    gd = D.add_gaussian_noise(d2, std=0.05, rng="device").cpu().numpy().astype(int)
    diff = gd - z["down2"].astype(int)
    assert 5.0 < diff.std() < 14.0 and abs(diff.mean()) < 1.0                            # sigma = 12.75 grey levels, clipped at the ends
    spd = D.add_salt_pepper_noise(d2, s=0.1, p=0.1, rng="device").cpu().numpy()
    changed = (spd != z["down2"]).any(-1).mean()
    assert 0.12 < changed < 0.25 and set(np.unique(spd[(spd != z["down2"]).any(-1)])) <= set(range(256))


def test_patch_batch_scaling_and_coords(dev):
    """dataset.patch_batch / PatchBank / GANDIV2KDataset.get_train_patches + scale_images (dataset.py:121-159): the float32
    batches equal the reference's expression sequence (ToTensor's /255, then /255 again, *2, -1) bit for bit, the patch
    positions follow its two randint draws, and the `reference_scaling=False` modes give [0,1] / [-1,1]."""
    DS = P("dataset")
    imgs_lr = [od.sample_image(f"in:pb_lr{i}", 24 + 4 * i, 40 + 8 * i) for i in range(3)]
    imgs_hr = [od.sample_image(f"in:pb_hr{i}", 4 * (24 + 4 * i), 4 * (40 + 8 * i)) for i in range(3)]
    pairs = [(torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)) for a, b in zip(imgs_lr, imgs_hr)]
    bank = DS.PatchBank(pairs, 4, (16, 8), rng=np.random.RandomState(11))
    idx = [2, 0, 1, 1, 2]
    lr, hr = bank.sample(5, indices=idx)
    assert tuple(lr.shape) == (5, 3, 8, 16) and tuple(hr.shape) == (5, 3, 32, 64) and lr.dtype == torch.float32
    rng = np.random.RandomState(11)
    for b, i in enumerate(idx):
        top, left, htop, hleft = od.train_patch_coords(imgs_lr[i].shape[0], imgs_lr[i].shape[1], 16, 8, 4, rng)
        rl, rh = od.scale_images(od.to_tensor(imgs_lr[i]), od.to_tensor(imgs_hr[i]))
        assert np.array_equal(lr[b].cpu().numpy(), rl[:, top:top + 8, left:left + 16])
        assert np.array_equal(hr[b].cpu().numpy(), rh[:, htop:htop + 32, hleft:hleft + 64])
    unit = DS.PatchBank(pairs, 4, (16, 8), reference_scaling=False, rng=np.random.RandomState(11)).sample(5, indices=idx)
    assert 0.0 <= float(unit[0].min()) and float(unit[0].max()) <= 1.0 and -1.0 <= float(unit[1].min()) and float(unit[1].max()) <= 1.0
    assert float((unit[0] - lr * 255.0).abs().max()) < 1e-6 and float((unit[1] - ((hr + 1) * 255.0 - 1)).abs().max()) < 1e-4
    # the Dataset-shaped surface: ToTensor, scale_images (in place, as written) and a patch cut as views
    ds = DS.GANDIV2KDataset.__new__(DS.GANDIV2KDataset)
    ds.LR_patch_size, ds.scale_factor = (16, 8), 4
    lt, ht = DS.to_tensor(pairs[0][0]), DS.to_tensor(pairs[0][1])
    assert np.array_equal(lt.cpu().numpy(), od.to_tensor(imgs_lr[0]))
    lt, ht = DS.GANDIV2KDataset.scale_images(lt, ht)
    rl, rh = od.scale_images(od.to_tensor(imgs_lr[0]), od.to_tensor(imgs_hr[0]))
    assert np.array_equal(lt.cpu().numpy(), rl) and np.array_equal(ht.cpu().numpy(), rh)
    np.random.seed(21)
    pl, ph_ = ds.get_train_patches(lt, ht)
    top, left, htop, hleft = od.train_patch_coords(24, 40, 16, 8, 4, np.random.RandomState(21))
    assert np.array_equal(pl.cpu().numpy(), rl[:, top:top + 8, left:left + 16])
    assert np.array_equal(ph_.cpu().numpy(), rh[:, htop:htop + 32, hleft:hleft + 64])
    with pytest.raises(RuntimeError):
        DS.patch_batch([pairs[0][0]], [20], [0], 8, 16, DS.PATCH_UNIT)                   # 20 + 8 > 24 rows: refused on the host


def test_get_image_pair_from_files(dev, tmp_path):
    """dataset.get_image_pair / GANDIV2KDataset.__getitem__ (dataset.py:9-62,161-171) on PNG files: the Pillow steps of the
    reference restated with Pillow on the host give the same tensors as the device path."""
    from PIL import Image
    DS = P("dataset")
    hr_dir, lr_dir = tmp_path / "HR", tmp_path / "LR"
    hr_dir.mkdir(), lr_dir.mkdir()
    hr_img = od.sample_image("in:gip_hr", 192, 256)
    lr_img = od.sample_image("in:gip_lr", 24, 32)
    Image.fromarray(hr_img).save(hr_dir / "0001.png")
    Image.fromarray(lr_img).save(lr_dir / "0001x8.png")
    ds = DS.GANDIV2KDataset(str(lr_dir), 4, downsample=False, noise_type=None, HR_dir=str(hr_dir), LR_patch_size=(8, 4), train=True,
                            device=dev)
    assert len(ds) == 1
    lr_t, hr_t, name = DS.get_image_pair(ds, 0)
    # the reference's lines with Pillow: both halved (dataset.py:22-23), HR resized to 4 x LR (:30-45)
    lr_p = Image.fromarray(lr_img).resize((16, 12), Image.BICUBIC)
    hr_p = Image.fromarray(hr_img).resize((128, 96), Image.BICUBIC).resize((64, 48), Image.BICUBIC)
    assert name == "0001"
    assert np.array_equal(lr_t.cpu().numpy(), od.to_tensor(np.array(lr_p)))
    assert np.array_equal(hr_t.cpu().numpy(), od.to_tensor(np.array(hr_p)))
    np.random.seed(4)
    pl, ph_, _ = ds[0]
    assert tuple(pl.shape) == (3, 4, 8) and tuple(ph_.shape) == (3, 16, 32) and pl.is_cuda
    assert float(pl.max()) <= 1.0 / 255.0 + 1e-9 and -1.0 <= float(ph_.min()) and float(ph_.max()) <= 2.0 / 255.0 - 1.0 + 1e-6
