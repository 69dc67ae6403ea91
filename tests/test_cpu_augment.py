"""CPU tier: the oracle restatement of the train-time transform chain against hand-derived answers."""
import numpy as np

from oracle import augment


def _src():
    # two 4x5 images whose pixel value encodes (image, y, x, c)
    s = np.zeros((2, 4, 5, 3), dtype=np.uint8)
    for i in range(2):
        for y in range(4):
            for x in range(5):
                for c in range(3):
                    s[i, y, x, c] = 100 * i + 20 * y + 4 * x + c
    return s


def test_crop_flip_semantics():
    s = _src()
    # no flip, crop 2x3 at (1, 2) of image 1
    o = augment.augment_batch(s, np.array([[1, 1, 2, 0]]), 2, 3, noise_std=0.0)
    want = s[1, 1:3, 2:5].astype(np.float32).transpose(2, 0, 1) / 255.0
    np.testing.assert_array_equal(o[0], want.astype(np.float32))
    # CIFAR order: flip the source, then crop: out[y, x] = src[top + y, Ws - 1 - (left + x)]
    o = augment.augment_batch(s, np.array([[0, 0, 1, 1]]), 2, 2, flip_first=True, noise_std=0.0)
    assert o[0, 0, 0, 0] == np.float32(s[0, 0, 5 - 1 - 1, 0]) / np.float32(255.0)
    assert o[0, 2, 1, 1] == np.float32(s[0, 1, 5 - 1 - 2, 2]) / np.float32(255.0)
    # CLIP order: crop, then flip the crop: out[y, x] = src[top + y, left + (Wo - 1 - x)]
    o = augment.augment_batch(s, np.array([[0, 0, 1, 1]]), 2, 2, flip_first=False, noise_std=0.0)
    assert o[0, 0, 0, 0] == np.float32(s[0, 0, 1 + 1, 0]) / np.float32(255.0)
    # RandomCrop(padding): negative origin reads zeros (fill 0)
    o = augment.augment_batch(s, np.array([[0, -1, -2, 0]]), 3, 4, noise_std=0.0)
    assert (o[0, :, 0, :] == 0).all() and (o[0, :, :, :2] == 0).all()
    assert o[0, 1, 1, 2] == np.float32(s[0, 0, 0, 1]) / np.float32(255.0)
    # Normalize
    o = augment.augment_batch(s, np.array([[1, 0, 0, 0]]), 4, 5, mean=[0.5, 0.25, 0.0], std=[0.5, 2.0, 1.0], noise_std=0.0)
    np.testing.assert_allclose(o[0, 1], (s[1, :, :, 1] / 255.0 - 0.25) / 2.0, rtol=1e-6)


def test_noise_generator_statistics_and_determinism():
    g = augment.noise(3, 8, 32, 32)
    assert g.shape == (8, 3, 32, 32) and np.isfinite(g).all()
    assert abs(g.mean()) < 0.02 and abs(g.std() - 1.0) < 0.02
    np.testing.assert_array_equal(g, augment.noise(3, 8, 32, 32))          # a pure function of (seed, slot, element)
    assert not np.array_equal(g, augment.noise(4, 8, 32, 32))
    assert abs(np.corrcoef(g[0].ravel(), g[1].ravel())[0, 1]) < 0.05       # slots are independent streams
    s = np.full((1, 32, 32, 3), 128, dtype=np.uint8)
    o = augment.augment_batch(s, np.array([[0, 0, 0, 0]]), 32, 32, noise_std=0.001, seed=3)
    d = o[0] - np.float32(128 / 255.0)
    assert abs(d.std() - 0.001) < 1e-4                                      # x + 0.001 * randn (train_cifar.py:36)
