"""GPU tier: the on-device input pipeline -- eoe_augment_batch against the oracle, the resident step-batch
source and a trainer run on it."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import augment as oaug   # noqa: E402


@pytest.mark.parametrize("Hs,crop,pad,flip_first", [(32, 32, 4, True), (40, 32, 0, False), (17, 9, 3, True)])
def test_augment_kernel_vs_oracle(Hs, crop, pad, flip_first):
    from eoe_amd.data import augment_batch
    rng = np.random.RandomState(0)
    src = rng.randint(0, 256, size=(11, Hs, Hs + 3, 3), dtype=np.uint8)
    n = 37
    params = np.stack([rng.randint(0, 11, n), rng.randint(-pad, Hs + pad - crop + 1, n), rng.randint(-pad, Hs + 3 + pad - crop + 1, n),
                       rng.randint(0, 2, n)], axis=1).astype(np.int32)
    mean, std = [0.49, 0.48, 0.45], [0.25, 0.24, 0.26]
    # without noise the chain is exact integer gathering + two fp32 ops: bit-exact
    got = augment_batch(torch.from_numpy(src).cuda(), torch.from_numpy(params).cuda(), (crop, crop), mean, std, flip_first, 0.0, 0)
    want = oaug.augment_batch(src, params, crop, crop, mean, std, flip_first, 0.0, 0)
    assert torch.equal(got.cpu(), torch.from_numpy(want))
    # with noise: the same counter-based draws (device logf / cosf vs numpy: a few ulp on a 0.001-scaled term)
    got = augment_batch(torch.from_numpy(src).cuda(), torch.from_numpy(params).cuda(), (crop, crop), mean, std, flip_first, 0.001, 5)
    want = oaug.augment_batch(src, params, crop, crop, mean, std, flip_first, 0.001, 5)
    assert (got.cpu() - torch.from_numpy(want)).abs().max().item() < 2e-6
    noise = (got.cpu().numpy() - oaug.augment_batch(src, params, crop, crop, mean, std, flip_first, 0.0, 0)) * np.array(std, dtype=np.float32).reshape(1, 3, 1, 1)
    assert abs(noise.std() - 0.001) < 5e-5 and abs(noise.mean()) < 2e-5


def test_resident_source_layout_and_trainer_run():
    """step batches keep the [normal half | OE half] layout with the OE index offset (bases.py:591-597); a CNN32 HSC run on
    the resident source learns to separate a bright OE set from a dark normal set"""
    from eoe_amd.data import ResidentImageSource
    from eoe_amd.models import CNN32
    from eoe_amd.training import TRAINER
    g = torch.Generator().manual_seed(0)
    normal = torch.randint(0, 128, (48, 32, 32, 3), generator=g, dtype=torch.uint8)
    oe = torch.randint(100, 256, (20, 32, 32, 3), generator=g, dtype=torch.uint8)
    test = torch.cat([torch.randint(0, 128, (16, 32, 32, 3), generator=g, dtype=torch.uint8),
                      torch.randint(100, 256, (16, 32, 32, 3), generator=g, dtype=torch.uint8)])
    ty = torch.cat([torch.zeros(16, dtype=torch.int64), torch.ones(16, dtype=torch.int64)])
    ds = ResidentImageSource(normal, oe, test, ty, crop=32, padding=4, mean=[0.35, 0.35, 0.35], std=[0.2, 0.2, 0.2], seed=1)
    train, tst = ds.loaders(16)
    batches = list(train)
    assert len(batches) == 3 and len(train) == 3
    for imgs, lbls, idcs in batches:
        n = lbls.shape[0] // 2
        assert imgs.is_cuda and imgs.shape == (2 * n, 3, 32, 32) and imgs.dtype == torch.float32
        assert lbls[:n].eq(0).all() and lbls[n:].eq(1).all()
        assert idcs[:n].max() < 48 and idcs[n:].min() >= 48 and idcs[n:].max() < 48 + 20
        assert imgs[:n].mean() < imgs[n:].mean()                 # dark normal half, bright OE half
    assert sorted(torch.cat([b[2][: b[1].shape[0] // 2] for b in batches]).tolist()) == list(range(48))   # every normal image once
    assert len(tst) == 2 and tst[0][0].shape == (16, 3, 32, 32)
    torch.manual_seed(0)
    tr = TRAINER["hsc"](CNN32(bias=True), dataset=ds, epochs=6, lr=1e-3, wdk=0.0, milestones=[], batch_size=16, classes=["dark"])
    _, res = tr.run(run_seeds=1)
    assert res["mean_auc"] > 0.95, res
    assert tr.last_losses[-1] < tr.last_losses[0]


def test_resize_and_color_jitter_equal_pillow(golden):
    """eoe_resize_pass_u8 / eoe_color_jitter_u8 / clip_preprocess against the bytes Pillow produced (fixture g14): the reference's
    Resize((256,256)) / Resize(256) / ColorJitter(0.01 x 4) / CLIP bicubic preprocessing run on PIL images (train_imagenet.py:31,
    train_clip_imagenet.py:28-29, train_cifar.py:32, clip.py:58-65) -- geometry and colour bit-exact, CLIP's floats to 1e-6"""
    import torch
    from eoe_amd import data
    g = golden("g14_pil_transforms")
    imgs = torch.from_numpy(g["images"]).cuda()
    for name, size, filt in (("bilinear_64x48", (64, 48), "bilinear"), ("bilinear_32x32", (32, 32), "bilinear"),
                             ("bilinear_150x200", (150, 200), "bilinear"), ("bicubic_64x48", (64, 48), "bicubic"),
                             ("bicubic_150x200", (150, 200), "bicubic"), ("bicubic_short56", 56, "bicubic")):
        got = data.resize_u8(imgs, size, filt).cpu().numpy()
        assert np.array_equal(got, g[f"resize/{name}"]), name
    assert data.resize_u8(imgs, (75, 100)).data_ptr() == imgs.data_ptr()              # nothing to do: both passes skipped
    jit = data.color_jitter_u8(imgs, torch.tensor([0, 1, 2]), torch.from_numpy(g["jitter/factors"]), torch.from_numpy(g["jitter/orders"]))
    assert np.array_equal(jit.cpu().numpy(), g["jitter/out"])
    # gather semantics + skipped ops: identity when every op code is out of range
    same = data.color_jitter_u8(imgs, torch.tensor([2, 0]), torch.ones(2, 4), torch.full((2, 4), 9, dtype=torch.int32))
    assert np.array_equal(same.cpu().numpy(), g["images"][[2, 0]])
    np.testing.assert_allclose(data.clip_preprocess(imgs, 32).cpu().numpy(), g["clip/out"], rtol=0, atol=1e-6)


def test_color_jitter_full_range_against_oracle():
    """every colour op at strong factors on random images, in all 24 orders, against oracle/augment.py (itself equal to Pillow)"""
    import itertools
    import torch
    from eoe_amd import data
    from oracle import augment, fill
    imgs = (fill.fill("cj/img", (24, 20, 28, 3), std=0.7) * 128 + 128).clip(0, 255).astype(np.uint8)
    orders = np.array(list(itertools.permutations(range(4))), np.int32)
    fac = np.stack([0.2 + 1.8 * fill.fill("cj/f", (24, 3), std=0.28).__abs__().clip(0, 1).reshape(24, 3)[:, i] for i in range(3)]
                   + [fill.fill("cj/h", (24,), std=0.25).clip(-0.5, 0.5)], axis=1).astype(np.float32)
    got = data.color_jitter_u8(torch.from_numpy(imgs).cuda(), torch.arange(24), torch.from_numpy(fac), torch.from_numpy(orders)).cpu().numpy()
    want = np.stack([augment.color_jitter(im, f, o) for im, f, o in zip(imgs, fac, orders)])
    assert np.array_equal(got, want), int(np.abs(got.astype(int) - want.astype(int)).max())


def test_resident_source_with_resize_and_jitter():
    """the ImageNet-style chain Resize((40,40)) -> ColorJitter -> RandomCrop(32) -> flip -> ToTensor -> noise -> Normalize through
    ResidentImageSource: shapes, label / index layout, finite values, and the same epoch twice from the same seed"""
    import torch
    from eoe_amd import data
    from oracle import fill
    mk = lambda tag, n: torch.from_numpy((fill.fill(tag, (n, 50, 60, 3), std=0.6) * 128 + 128).clip(0, 255).astype(np.uint8))   # noqa: E731
    def epoch():
        src = data.ResidentImageSource(mk("rs/n", 10), mk("rs/o", 6), mk("rs/t", 4), torch.tensor([0, 0, 1, 1]), crop=32, resize=(40, 40),
                                       test_resize=(40, 40), color_jitter=(0.01, 0.01, 0.01, 0.01), mean=(0.5, 0.5, 0.5),
                                       std=(0.25, 0.25, 0.25), flip_first=False, seed=3)
        train, test = src.loaders(4)
        return [(x.cpu(), y, i) for x, y, i in train], test
    a, test = epoch()
    b, _ = epoch()
    assert len(a) == 3 and a[0][0].shape == (8, 3, 32, 32) and a[2][0].shape == (4, 3, 32, 32)
    assert a[0][1].tolist() == [0] * 4 + [1] * 4 and all(int(i) >= 10 for i in a[0][2][4:])
    assert all(torch.isfinite(x).all() for x, _, _ in a) and test[0][0].shape == (4, 3, 32, 32)
    for (xa, _, ia), (xb, _, ib) in zip(a, b):
        assert torch.equal(xa, xb) and torch.equal(ia, ib)
