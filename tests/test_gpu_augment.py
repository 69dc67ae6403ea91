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
