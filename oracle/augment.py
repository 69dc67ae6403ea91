"""CPU restatement of the train-time input transform chain (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py).

Follows the reference's runners: `src/eoe/main/train_cifar.py:31-38` (RandomHorizontalFlip -> RandomCrop(32, padding=4) ->
ToTensor -> x + 0.001 * randn_like(x) -> 'normalize') and `src/eoe/main/train_clip_imagenet.py:27-36` (RandomCrop(224) ->
RandomHorizontalFlip -> ... ToTensor -> noise -> normalize), with the device-side Normalize of
`src/eoe/training/ad_trainer.py:413-425` (`utils/transformations.py:126-138`: (x - mean) / std per channel).
torchvision semantics restated: RandomCrop pads with zeros and cuts [top, top+S) x [left, left+S) of the padded image
(here top/left are given relative to the UNPADDED image, so they may be negative); hflip reverses the width axis;
ToTensor divides uint8 by 255.  ColorJitter(0.01) is outside SURVEY.md row N1.

The random draws (crop origin, flip) are inputs; the noise is the counter-based rule of `eoe_augment_batch`
(include/eoe_hip.h): element e = (c*Ho + y)*Wo + x of batch slot b draws a Box-Muller normal from
splitmix64(seed*2^40 + b*2^18 + e) -- torch's randn stream is not portable, the distribution is what is reproduced.
"""
import numpy as np

from .fill import _splitmix64


def noise(seed: int, n: int, Ho: int, Wo: int) -> np.ndarray:
    """standard-normal float32 [n, 3, Ho, Wo] of the counter-based generator"""
    e = np.arange(3 * Ho * Wo, dtype=np.uint64)[None, :]
    b = np.arange(n, dtype=np.uint64)[:, None]
    ctr = (np.uint64(seed) << np.uint64(40)) + (b << np.uint64(18)) + e
    z = _splitmix64(ctr)
    u1 = ((z >> np.uint64(40)) + np.uint64(1)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    u2 = ((z >> np.uint64(16)) & np.uint64(0xFFFFFF)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    g = np.sqrt(np.float32(-2.0) * np.log(u1)) * np.cos(np.float32(6.283185307179586) * u2)
    return g.astype(np.float32).reshape(n, 3, Ho, Wo)


def augment_batch(src: np.ndarray, params: np.ndarray, Ho: int, Wo: int, mean=None, std=None, flip_first: bool = True,
                  noise_std: float = 0.001, seed: int = 0) -> np.ndarray:
    """src uint8 [n_src, Hs, Ws, 3]; params int [n, 4] = (index, top, left, flip) -> float32 NCHW [n, 3, Ho, Wo]"""
    n = params.shape[0]
    _, Hs, Ws, _ = src.shape
    out = np.zeros((n, 3, Ho, Wo), dtype=np.float32)
    for b in range(n):
        idx, top, left, flip = (int(v) for v in params[b])
        img = src[idx]                                          # [Hs, Ws, 3]
        if flip_first and flip:
            img = img[:, ::-1]                                  # hflip the source (train_cifar.py:33)
        ys, xs = np.arange(Ho) + top, np.arange(Wo) + left
        crop = np.zeros((Ho, Wo, 3), dtype=np.float32)          # RandomCrop padding: fill 0
        vy, vx = (ys >= 0) & (ys < Hs), (xs >= 0) & (xs < Ws)
        crop[np.ix_(vy, vx)] = img[np.ix_(ys[vy], xs[vx])]
        if (not flip_first) and flip:
            crop = crop[:, ::-1]                                # hflip the crop (train_clip_imagenet.py:31)
        out[b] = (crop / np.float32(255.0)).transpose(2, 0, 1)  # ToTensor
    if noise_std > 0:
        out = out + np.float32(noise_std) * noise(seed, n, Ho, Wo)
    if mean is not None:
        m = np.asarray(mean, dtype=np.float32).reshape(1, 3, 1, 1)
        s = np.asarray(std, dtype=np.float32).reshape(1, 3, 1, 1)
        out = (out - m) / s
    return out.astype(np.float32)
