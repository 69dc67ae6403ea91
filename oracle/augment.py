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


# ----------------------------------------------------------------------------------------------------------------------------
# Resize and ColorJitter of the reference's runners (`main/train_imagenet.py:30-31` Resize(256); `main/train_clip_imagenet.py:28-29`
# Resize((256, 256)) + ColorJitter(0.01 x 4); `main/train_cifar.py:32`; CLIP's `_transform`, `clip_official/clip/clip.py:58-65`:
# Resize(224, bicubic) -> CenterCrop(224) -> ToTensor -> Normalize(CLIP mean / std)).  These run on PIL images in the reference:
# torchvision (>= 0.18.1, `src/requirements.txt`; absent in this container) hands them to Pillow, whose published algorithms are
# restated here in integer arithmetic and pinned against Pillow itself (present: 12.2.0) by tests/golden fixture g14:
#   * Image.resize (libImaging/Resample.c): separable, horizontal pass then vertical pass on uint8, the filter stretched by the
#     down-scaling factor (antialias), weights normalised in double and rounded to 22-bit fixed point, 0.5 rounding offset;
#   * ImageEnhance.Brightness / Contrast / Color = Image.blend(degenerate, image, factor) (libImaging/Blend.c: truncation inside
#     [0, 1], clipped truncation outside), degenerates: black / the rounded mean of the L image / the L image;
#   * L = (19595 R + 38470 G + 7471 B + 0x8000) >> 16 (libImaging/Convert.c);
#   * hue: RGB -> HSV -> h += uint8(hue_factor * 255) (mod 256) -> RGB, Pillow's 8-bit HSV (Convert.c rgb2hsv / hsv2rgb).
# ----------------------------------------------------------------------------------------------------------------------------
PRECISION_BITS = 32 - 8 - 2


def _bilinear(x):
    x = abs(x)
    return 1.0 - x if x < 1.0 else 0.0


def _bicubic(x, a=-0.5):
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


FILTERS = {"bilinear": (_bilinear, 1.0), "bicubic": (_bicubic, 2.0)}


def resize_coeffs(in_size: int, out_size: int, filt: str):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc: per output index the first source index, the tap count and
    the fixed-point weights (int32 [out_size, ksize])"""
    fn, support0 = FILTERS[filt]
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = support0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = np.array([fn((x + xmin - center + 0.5) * ss) for x in range(xmax)], dtype=np.float64)
        ww = w.sum()
        if ww != 0.0:
            w = w / ww
        for x in range(xmax):
            kk[xx, x] = int(w[x] * (1 << PRECISION_BITS) - 0.5) if w[x] < 0 else int(w[x] * (1 << PRECISION_BITS) + 0.5)
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _resample_axis(img: np.ndarray, out_size: int, filt: str, axis: int) -> np.ndarray:
    bounds, kk = resize_coeffs(img.shape[axis], out_size, filt)
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((out_size,) + src.shape[1:], dtype=np.uint8)
    for xx in range(out_size):
        xmin, n = bounds[xx]
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), dtype=np.int64)
        for x in range(n):
            acc += src[xmin + x] * int(kk[xx, x])
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize(img: np.ndarray, size, filt: str = "bilinear") -> np.ndarray:
    """uint8 [H, W, 3] -> uint8 [Ho, Wo, 3] as `torchvision.transforms.Resize(size)` does on a PIL image: a pair (h, w), or
    an int = the shorter side (the longer one int(size * long / short), torchvision's rule); horizontal pass first, a pass
    whose size does not change is skipped (Resample.c ImagingResample)"""
    H, W = img.shape[:2]
    if isinstance(size, int):
        if W <= H:
            Wo, Ho = size, int(size * H / W)
        else:
            Ho, Wo = size, int(size * W / H)
    else:
        Ho, Wo = size
    if Wo != W:
        img = _resample_axis(img, Wo, filt, 1)
    if Ho != H:
        img = _resample_axis(img, Ho, filt, 0)
    return img


def to_gray(img: np.ndarray) -> np.ndarray:
    r, g, b = (img[..., i].astype(np.int64) for i in range(3))
    return ((r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16).astype(np.uint8)


def blend(degenerate: np.ndarray, img: np.ndarray, alpha: float) -> np.ndarray:
    """Blend.c: out = in1 + alpha * (in2 - in1) in float; truncated for alpha in [0, 1], clipped then truncated outside"""
    a = np.float32(alpha)
    t = degenerate.astype(np.float32) + a * (img.astype(np.float32) - degenerate.astype(np.float32))
    if 0.0 <= alpha <= 1.0:
        return t.astype(np.int64).astype(np.uint8)
    return np.where(t <= 0.0, 0, np.where(t >= 255.0, 255, t.astype(np.int64))).astype(np.uint8)


def adjust_brightness(img, f):
    return blend(np.zeros_like(img), img, f)


def adjust_contrast(img, f):
    mean = int(to_gray(img).astype(np.float64).mean() + 0.5)             # ImageEnhance.Contrast: int(ImageStat.Stat(L).mean[0] + 0.5)
    return blend(np.full_like(img, mean), img, f)


def adjust_saturation(img, f):
    return blend(np.repeat(to_gray(img)[..., None], 3, axis=-1), img, f)


def rgb_to_hsv_u8(img: np.ndarray) -> np.ndarray:
    r, g, b = (img[..., i].astype(np.int64) for i in range(3))
    maxc, minc = np.maximum(np.maximum(r, g), b), np.minimum(np.minimum(r, g), b)
    cr = (maxc - minc).astype(np.float32)
    safe = np.where(cr == 0, np.float32(1), cr)
    s = cr / np.where(maxc == 0, 1, maxc).astype(np.float32)
    rc, gc, bc = ((maxc - c).astype(np.float32) / safe for c in (r, g, b))
    # C semantics of Convert.c: `bc - gc` is a float operation; `2.0 + rc - bc` and `h / 6.0 + 1.0` are evaluated in double
    # (double literals) and rounded once when stored to the float h
    rc64, gc64, bc64 = rc.astype(np.float64), gc.astype(np.float64), bc.astype(np.float64)
    h = np.where(r == maxc, (bc - gc).astype(np.float64), np.where(g == maxc, 2.0 + rc64 - bc64, 4.0 + gc64 - rc64)).astype(np.float32)
    h = np.fmod(h.astype(np.float64) / 6.0 + 1.0, 1.0).astype(np.float32).astype(np.float64)
    uh = np.clip((h * 255.0).astype(np.int64), 0, 255)
    us = np.clip((s.astype(np.float64) * 255.0).astype(np.int64), 0, 255)
    gray = maxc == minc
    return np.stack([np.where(gray, 0, uh), np.where(gray, 0, us), maxc], axis=-1).astype(np.uint8)


def hsv_to_rgb_u8(hsv: np.ndarray) -> np.ndarray:
    h, s, v = (hsv[..., i].astype(np.int64) for i in range(3))
    hf = h.astype(np.float32) * np.float32(6.0) / np.float32(255.0)
    i = np.floor(hf).astype(np.int64)
    f = hf - i.astype(np.float32)
    fs = s.astype(np.float32) / np.float32(255.0)
    vf = v.astype(np.float32)
    rnd = lambda x: np.floor(x.astype(np.float64) + 0.5).astype(np.int64)      # noqa: E731  (C round() on non-negative values)
    p = rnd(vf * (np.float32(1.0) - fs))
    q = rnd(vf * (np.float32(1.0) - fs * f))
    t = rnd(vf * (np.float32(1.0) - fs * (np.float32(1.0) - f)))
    sel = i % 6
    r = np.choose(sel, [v, q, p, p, t, v])
    g = np.choose(sel, [t, v, v, q, p, p])
    b = np.choose(sel, [p, p, t, v, v, q])
    out = np.stack([r, g, b], axis=-1)
    out = np.where((s == 0)[..., None], v[..., None], out)
    return np.clip(out, 0, 255).astype(np.uint8)


def adjust_hue(img, hue_factor):
    hsv = rgb_to_hsv_u8(img)
    shift = np.uint8(int(hue_factor * 255) & 0xFF)                          # torchvision: np_h += np.uint8(hue_factor * 255)
    hsv[..., 0] = (hsv[..., 0].astype(np.int64) + int(shift)) & 0xFF
    return hsv_to_rgb_u8(hsv)


JITTER_OPS = (adjust_brightness, adjust_contrast, adjust_saturation, adjust_hue)


def color_jitter(img: np.ndarray, factors, order) -> np.ndarray:
    """torchvision.transforms.ColorJitter.forward with its random draws made explicit: `order` is the permutation of
    (0 brightness, 1 contrast, 2 saturation, 3 hue), `factors` = (b, c, s, h) as sampled (b, c, s around 1; h around 0)"""
    for op in order:
        img = JITTER_OPS[int(op)](img, float(factors[int(op)]))
    return img
