"""The step-batch contract of the hot path and synthetic sources that honour it.

The reference's datasets / PIL pipelines are out of scope (disk I/O; absent in both containers); what the
training loop depends on is the layout `BalancedConcatLoader` produces (`src/eoe/datasets/bases.py:570-600`):
  imgs   = cat([normal_i, oe_i[:len(normal_i)]])        labels = [nominal]*n + [anomalous]*n
  idcs   = cat([normal idcs, oe idcs + len(normal dataset)])
with OE indices tiled when the OE set is smaller than the normal set (:580-584), OE sampled with replacement iff it
holds >= 10 000 samples (`bases.py:561`), a ragged last batch (no drop_last) and len(loader) = len(normal loader).
"""
import math
from typing import Iterator, List, Optional, Sequence, Tuple

import torch


def balanced_concat(normal: Sequence[torch.Tensor], oe_iter: Iterator, n_normal_dataset: int) -> List[torch.Tensor]:
    """BalancedConcatLoader.__next__ (bases.py:591-597) on (imgs, lbls, idcs) triples"""
    oe = [a for a in next(oe_iter)]
    while oe[1].shape[0] < normal[1].shape[0]:
        oe = [torch.cat([a, b]) for a, b in zip(oe, next(oe_iter))]
    oe[-1] = oe[-1] + n_normal_dataset
    n = normal[0].shape[0]
    return [torch.cat([i, j[:n]]) for i, j in zip(normal, oe)]


def tile_oe_indices(oe_indices: torch.Tensor, n_normal: int) -> torch.Tensor:
    if len(oe_indices) < n_normal:
        r = int(math.ceil(n_normal / len(oe_indices)))
        oe_indices = oe_indices.reshape(1, -1).repeat(r, 1).reshape(-1)
    return oe_indices


class ListSource:
    """a fixed list of step batches [(imgs, lbls[, idcs]), ...] per epoch -- what the parity tests feed"""

    nominal_label, anomalous_label = 0, 1

    def __init__(self, train_batches, test_batches=None, normalize=None):
        self.train_batches, self.test_batches, self.normalize = list(train_batches), list(test_batches or []), normalize
        self.ds_statistics = None

    def loaders(self, batch_size=None, **kw):
        return self.train_batches, self.test_batches


class SyntheticAD:
    """in-memory synthetic one-vs-rest task: normal samples ~ N(0,1), anomalies / OE ~ N(0,1) + shift * pattern.
    Produces step batches with the BalancedConcatLoader layout; test split holds labelled normal + anomalous."""

    nominal_label, anomalous_label = 0, 1

    def __init__(self, n_train_normal=512, n_oe=512, n_test=256, res=224, shift=0.5, seed=0, normalize=None):
        g = torch.Generator().manual_seed(seed)
        self.res = res
        pattern = torch.randn((1, 3, res, res), generator=g)
        self.train_normal = torch.randn((n_train_normal, 3, res, res), generator=g)
        self.oe = torch.randn((n_oe, 3, res, res), generator=g) + shift * pattern
        half = n_test // 2
        self.test_x = torch.cat([torch.randn((half, 3, res, res), generator=g),
                                 torch.randn((n_test - half, 3, res, res), generator=g) + shift * pattern])
        self.test_y = torch.cat([torch.zeros(half, dtype=torch.int64), torch.ones(n_test - half, dtype=torch.int64)])
        self.normalize = normalize
        self.ds_statistics = None
        self._g = g

    def _epoch(self, batch_size):
        n, m = self.train_normal.shape[0], self.oe.shape[0]
        perm = torch.randperm(n, generator=self._g)
        oe_idx = tile_oe_indices(torch.arange(m), n)
        if m >= 10000:                                   # bases.py:561: with replacement for large OE sets
            oe_order = oe_idx[torch.randint(len(oe_idx), (len(oe_idx),), generator=self._g)]
        else:
            oe_order = oe_idx[torch.randperm(len(oe_idx), generator=self._g)]

        def oe_batches():
            for s in range(0, len(oe_order), batch_size):
                idx = oe_order[s:s + batch_size]
                yield self.oe[idx], torch.ones(len(idx), dtype=torch.int64), idx.clone()

        oe_it = oe_batches()
        for s in range(0, n, batch_size):
            idx = perm[s:s + batch_size]
            normal = (self.train_normal[idx], torch.zeros(len(idx), dtype=torch.int64), idx.clone())
            yield tuple(balanced_concat(normal, oe_it, n))

    def loaders(self, batch_size, **kw):
        class _Train:
            def __init__(s, outer):
                s.outer = outer

            def __iter__(s):
                return s.outer._epoch(batch_size)

            def __len__(s):
                return math.ceil(s.outer.train_normal.shape[0] / batch_size)

        test = [(self.test_x[s:s + batch_size], self.test_y[s:s + batch_size],
                 torch.arange(s, min(s + batch_size, len(self.test_y)))) for s in range(0, len(self.test_y), batch_size)]
        return _Train(self), test



# ---------------------------------------------------------------------------------------------------------------------
# on-device input pipeline
# ---------------------------------------------------------------------------------------------------------------------
def augment_batch(src_u8, params, out_hw, mean=None, std=None, flip_first=True, noise_std=0.001, seed=0):
    """gather + RandomCrop(zero padding) + RandomHorizontalFlip + ToTensor + noise + Normalize in ONE kernel over a uint8
    NHWC image set resident in HBM (`eoe_augment_batch`, include/eoe_hip.h): replaces the PIL transform chain of
    `main/train_cifar.py:31-38` / `main/train_clip_imagenet.py:27-36` and the Normalize of `ad_trainer.py:413-425`.
    src_u8 uint8 [n_src,Hs,Ws,3] (GPU); params int32 [n,4] = (index, top, left, flip) (GPU) -> fp32 NCHW [n,3,Ho,Wo]"""
    import ctypes as C                                  # noqa: F401
    from ._lib import check, lib
    if not (src_u8.is_cuda and params.is_cuda):
        raise RuntimeError("augment_batch needs GPU tensors (there is no CPU fallback)")
    assert src_u8.dtype == torch.uint8 and src_u8.dim() == 4 and src_u8.shape[3] == 3 and src_u8.is_contiguous()
    assert params.dtype == torch.int32 and params.dim() == 2 and params.shape[1] == 4 and params.is_contiguous()
    n, (Ho, Wo) = params.shape[0], out_hw
    dev = src_u8.device
    m = torch.as_tensor(mean, dtype=torch.float32, device=dev).contiguous() if mean is not None else None
    s = torch.as_tensor(std, dtype=torch.float32, device=dev).contiguous() if std is not None else None
    out = torch.empty((n, 3, Ho, Wo), dtype=torch.float32, device=dev)
    check(lib.eoe_augment_batch(src_u8.data_ptr(), src_u8.shape[0], src_u8.shape[1], src_u8.shape[2], params.data_ptr(),
                                None if m is None else m.data_ptr(), None if s is None else s.data_ptr(), out.data_ptr(), n, Ho, Wo,
                                1 if flip_first else 0, float(noise_std), int(seed), torch.cuda.current_stream().cuda_stream),
          "eoe_augment_batch")
    return out


def _resize_tables(in_size, out_size, filt, device):
    """Pillow's filter taps for one axis from the library's host helper, uploaded once"""
    import ctypes as C
    from ._lib import check, lib
    k = C.c_int(0)
    check(lib.eoe_resize_coeffs(in_size, out_size, filt, None, None, 0, C.byref(k)), "eoe_resize_coeffs")
    bounds = torch.empty((out_size, 2), dtype=torch.int32)
    kk = torch.empty((out_size, k.value), dtype=torch.int32)
    check(lib.eoe_resize_coeffs(in_size, out_size, filt, bounds.data_ptr(), kk.data_ptr(), k.value, None), "eoe_resize_coeffs")
    return bounds.to(device), kk.to(device), k.value


def resize_u8(src_u8, size, interpolation="bilinear"):
    """`torchvision.transforms.Resize(size)` as the reference applies it to PIL images (`main/train_imagenet.py:31`,
    `main/train_clip_imagenet.py:28`; bicubic for CLIP's preprocessing, `clip_official/clip/clip.py:60`), on a uint8 NHWC image
    set in HBM and byte-exact with Pillow: `size` = (h, w), or an int = the shorter side (the other int(size * long / short)).
    Deterministic, so a resident dataset is resized ONCE, not per step."""
    from ._lib import check, lib, EOE_RESIZE_BILINEAR, EOE_RESIZE_BICUBIC
    if not src_u8.is_cuda:
        raise RuntimeError("resize_u8 needs a GPU tensor (there is no CPU fallback)")
    assert src_u8.dtype == torch.uint8 and src_u8.dim() == 4 and src_u8.shape[3] == 3 and src_u8.is_contiguous()
    filt = {"bilinear": EOE_RESIZE_BILINEAR, "bicubic": EOE_RESIZE_BICUBIC}[interpolation]
    n, H, W, _ = src_u8.shape
    if isinstance(size, int):
        Ho, Wo = (int(size * H / W), size) if W <= H else (size, int(size * W / H))
    else:
        Ho, Wo = size
    st = torch.cuda.current_stream().cuda_stream
    cur = src_u8
    if Wo != W:
        b, k, ks = _resize_tables(W, Wo, filt, src_u8.device)
        nxt = torch.empty((n, H, Wo, 3), dtype=torch.uint8, device=src_u8.device)
        check(lib.eoe_resize_pass_u8(cur.data_ptr(), nxt.data_ptr(), b.data_ptr(), k.data_ptr(), ks, n * H, W, Wo, 3, st), "eoe_resize_pass_u8")
        cur = nxt
    if Ho != H:
        b, k, ks = _resize_tables(H, Ho, filt, src_u8.device)
        nxt = torch.empty((n, Ho, Wo, 3), dtype=torch.uint8, device=src_u8.device)
        check(lib.eoe_resize_pass_u8(cur.data_ptr(), nxt.data_ptr(), b.data_ptr(), k.data_ptr(), ks, n, H, Ho, Wo * 3, st), "eoe_resize_pass_u8")
        cur = nxt
    return cur


def color_jitter_u8(src_u8, idx, factors, order):
    """`torchvision.transforms.ColorJitter` with its random draws made explicit (`main/train_cifar.py:32`,
    `main/train_clip_imagenet.py:29`: brightness = contrast = saturation = hue = 0.01), byte-exact with Pillow: gathers
    src_u8[idx] (uint8 NHWC) and applies, per image, the four ops in `order` (int32 [n, 4], a permutation of 0 brightness,
    1 contrast, 2 saturation, 3 hue) with `factors` (fp32 [n, 4] = b, c, s around 1 and h around 0) -> uint8 [n, H, W, 3]"""
    from ._lib import check, lib
    if not src_u8.is_cuda:
        raise RuntimeError("color_jitter_u8 needs GPU tensors (there is no CPU fallback)")
    dev = src_u8.device
    idx = idx.to(device=dev, dtype=torch.int32).contiguous()
    factors = factors.to(device=dev, dtype=torch.float32).contiguous()
    order = order.to(device=dev, dtype=torch.int32).contiguous()
    n, (_, H, W, _) = idx.shape[0], src_u8.shape
    assert factors.shape == (n, 4) and order.shape == (n, 4) and src_u8.dtype == torch.uint8 and src_u8.is_contiguous()
    out = torch.empty((n, H, W, 3), dtype=torch.uint8, device=dev)
    scratch = torch.empty(n, dtype=torch.int32, device=dev)
    check(lib.eoe_color_jitter_u8(src_u8.data_ptr(), src_u8.shape[0], idx.data_ptr(), factors.data_ptr(), order.data_ptr(),
                                  scratch.data_ptr(), out.data_ptr(), n, H, W, torch.cuda.current_stream().cuda_stream), "eoe_color_jitter_u8")
    return out


def sample_color_jitter(n, brightness, contrast, saturation, hue, generator=None):
    """the draws of `ColorJitter.get_params`: a random permutation of the four ops and uniform factors in
    [max(0, 1 - x), 1 + x] (hue: [-x, x]) per image"""
    order = torch.stack([torch.randperm(4, generator=generator) for _ in range(n)]).to(torch.int32)
    u = torch.rand((n, 4), generator=generator)
    lo = torch.tensor([max(0.0, 1 - brightness), max(0.0, 1 - contrast), max(0.0, 1 - saturation), -hue])
    hi = torch.tensor([1 + brightness, 1 + contrast, 1 + saturation, hue])
    return (lo + u * (hi - lo)).to(torch.float32), order


CLIP_MEAN, CLIP_STD = (0.48145466, 0.4578275, 0.40821073), (0.26862954, 0.26130258, 0.27577711)      # clip.py:64


def clip_preprocess(src_u8, n_px=224):
    """CLIP's `_transform` (`clip_official/clip/clip.py:58-65`): Resize(n_px, bicubic) -> CenterCrop(n_px) -> ToTensor ->
    Normalize(CLIP mean / std), on a uint8 NHWC set in HBM -> fp32 NCHW"""
    r = resize_u8(src_u8, n_px, "bicubic")
    n, H, W, _ = r.shape
    # torchvision's CenterCrop: top = int(round((H - n_px) / 2.0))
    idx = torch.arange(n)
    p = torch.stack([idx, torch.full_like(idx, int(round((H - n_px) / 2.0))), torch.full_like(idx, int(round((W - n_px) / 2.0))),
                     torch.zeros_like(idx)], dim=1).to(torch.int32).to(r.device)
    return augment_batch(r, p, (n_px, n_px), CLIP_MEAN, CLIP_STD, True, 0.0, 0)


class ResidentImageSource:
    """step-batch source whose uint8 images live in HBM: every step batch ([normal half | OE half], the BalancedConcatLoader
    contract of `datasets/bases.py:570-600`) is gathered, cropped, flipped, noised and normalised by one kernel; the host only
    draws (index, crop origin, flip) per sample.  Batches come out already normalised, so `.normalize` is None (the trainer
    then installs no second Normalize).

    `normal_index` (optional): the rows of `normal_u8` that ARE the normal training set -- the reference's `Subset` over the
    samples of the normal classes (`bases.py:169-203`); batches report those rows' indices in the full set, as the reference's
    datasets do (`cifar.py:106-121`), and OE indices are offset by the length of the FULL normal set (`bases.py:596`)."""

    nominal_label, anomalous_label = 0, 1

    def __init__(self, normal_u8, oe_u8, test_u8, test_labels, crop, padding=0, mean=None, std=None, flip_first=True,
                 noise_std=0.001, seed=0, device="cuda", resize=None, test_resize=None, color_jitter=None, interpolation="bilinear",
                 normal_index=None):
        """resize / test_resize: `transforms.Resize` argument applied once to the resident train / test sets (None: as given);
        color_jitter: (brightness, contrast, saturation, hue) of `transforms.ColorJitter`, drawn per sample per step"""
        dev = torch.device(device)
        self.normal, self.oe, self.test = (t.to(dev).contiguous() for t in (normal_u8, oe_u8, test_u8))
        if resize is not None:
            self.normal, self.oe = resize_u8(self.normal, resize, interpolation), resize_u8(self.oe, resize, interpolation)
        if test_resize is not None:
            self.test = resize_u8(self.test, test_resize, interpolation)
        self.color_jitter = color_jitter
        self.test_y = test_labels.clone()
        self.crop, self.padding, self.mean, self.std = int(crop), int(padding), mean, std
        self.flip_first, self.noise_std, self.seed = flip_first, noise_std, int(seed)
        self.normalize = None
        self.ds_statistics = None
        self.normal_index = None if normal_index is None else torch.as_tensor(normal_index, dtype=torch.int64).clone()
        self._g = torch.Generator().manual_seed(seed)
        self._step = 0

    def _params(self, idx, Hs, Ws):
        n = len(idx)
        top = torch.randint(-self.padding, Hs + self.padding - self.crop + 1, (n,), generator=self._g)
        left = torch.randint(-self.padding, Ws + self.padding - self.crop + 1, (n,), generator=self._g)
        flip = torch.randint(0, 2, (n,), generator=self._g)
        return torch.stack([idx.to(torch.int64), top, left, flip], dim=1).to(torch.int32)

    def _epoch(self, batch_size):
        subset = self.normal_index if self.normal_index is not None else torch.arange(self.normal.shape[0])
        n, n_full, m = len(subset), self.normal.shape[0], self.oe.shape[0]
        perm = subset[torch.randperm(n, generator=self._g)]
        oe_idx = tile_oe_indices(torch.arange(m), n)
        if m >= 10000:                                   # bases.py:561: OE sets of >= 10 000 samples are drawn with replacement
            oe_order = oe_idx[torch.randint(len(oe_idx), (len(oe_idx),), generator=self._g)]
        else:
            oe_order = oe_idx[torch.randperm(len(oe_idx), generator=self._g)]
        dev = self.normal.device
        for s in range(0, n, batch_size):
            ni = perm[s:s + batch_size]
            oi = oe_order[s:s + batch_size][:len(ni)]            # the OE half is cut to the normal half's size (bases.py:597)
            self._step += 1
            # two launches (normal half from its image set, OE half from the other), written into one batch tensor
            pn = self._params(ni, self.normal.shape[1], self.normal.shape[2]).to(dev)
            po = self._params(oi, self.oe.shape[1], self.oe.shape[2]).to(dev)
            seed = (self.seed * 65521 + self._step) % (1 << 23)
            src_n, src_o = self.normal, self.oe
            if self.color_jitter is not None:
                # ColorJitter comes first in the reference's chains (train_cifar.py:32, train_clip_imagenet.py:29): the gathered,
                # jittered uint8 images become the "set" the crop / flip kernel reads (slot i = image i)
                fn, on = sample_color_jitter(len(ni), *self.color_jitter, generator=self._g)
                fo, oo = sample_color_jitter(len(oi), *self.color_jitter, generator=self._g)
                src_n, src_o = color_jitter_u8(self.normal, ni, fn, on), color_jitter_u8(self.oe, oi, fo, oo)
                pn[:, 0] = torch.arange(len(ni), dtype=torch.int32, device=dev)
                po[:, 0] = torch.arange(len(oi), dtype=torch.int32, device=dev)
            xn = augment_batch(src_n, pn, (self.crop, self.crop), self.mean, self.std, self.flip_first, self.noise_std, 2 * seed)
            xo = augment_batch(src_o, po, (self.crop, self.crop), self.mean, self.std, self.flip_first, self.noise_std, 2 * seed + 1)
            lbls = torch.cat([torch.full((len(ni),), self.nominal_label, dtype=torch.int64),
                              torch.full((len(oi),), self.anomalous_label, dtype=torch.int64)])
            yield torch.cat([xn, xo]), lbls, torch.cat([ni, oi + n_full])      # OE indices offset by the FULL normal set (bases.py:596)

    def loaders(self, batch_size, **kw):
        outer = self

        class _Train:
            def __iter__(s):
                return outer._epoch(batch_size)

            def __len__(s):
                n = outer.normal.shape[0] if outer.normal_index is None else len(outer.normal_index)
                return math.ceil(n / batch_size)

        # test split: centre crop, no flip, no noise (val_transform: ToTensor + normalize, train_cifar.py:39-42)
        Hs, Ws = self.test.shape[1], self.test.shape[2]
        test = []
        for s in range(0, len(self.test_y), batch_size):
            idx = torch.arange(s, min(s + batch_size, len(self.test_y)))
            p = torch.stack([idx, torch.full_like(idx, (Hs - self.crop) // 2), torch.full_like(idx, (Ws - self.crop) // 2),
                             torch.zeros_like(idx)], dim=1).to(torch.int32).to(self.test.device)
            test.append((augment_batch(self.test, p, (self.crop, self.crop), self.mean, self.std, True, 0.0, 0), self.test_y[idx], idx))
        return _Train(), test


def normal_subset(class_labels, normal_classes) -> torch.Tensor:
    """rows of a labelled set that belong to the normal classes, ascending: `TorchvisionDataset.create_subset`
    (`datasets/bases.py:192-195`: np.argwhere(np.isin(labels, normal_classes)))"""
    lab = torch.as_tensor(class_labels, dtype=torch.int64)
    keep = torch.zeros(len(lab), dtype=torch.bool)
    for c in normal_classes:
        keep |= lab == int(c)
    return torch.nonzero(keep).flatten()


def ad_targets(class_labels, normal_classes, nominal_label: int = 0) -> torch.Tensor:
    """the reference's `target_transform` (`datasets/bases.py:137-139`): anomalous iff the sample's class is not a normal class"""
    lab = torch.as_tensor(class_labels, dtype=torch.int64)
    normal = torch.zeros(len(lab), dtype=torch.bool)
    for c in normal_classes:
        normal |= lab == int(c)
    return torch.where(normal, torch.tensor(nominal_label), torch.tensor(1 - nominal_label)).to(torch.int64)


class LabelledImageSet:
    """A multi-class image set resident in HBM (uint8 NHWC + integer class labels) from which the class x seed loop of
    `ADTrainer.run` draws one anomaly-detection task per call, as the reference's `load_dataset(dsstr, datapath,
    self.get_nominal_classes(c), 0, ...)` does (`training/ad_trainer.py:248-253`, `datasets/__init__.py:237-340`):
      * normal training samples = the training rows whose class is one of `normal_classes` (`bases.py:169-203`);
      * the test split is the WHOLE test set, labelled nominal (0) for the normal classes and anomalous (1) for the rest
        (`bases.py:130-139`) -- under `leave_one_out` the anomalies are the one held-out class, under `one_vs_rest` all others;
      * outlier exposure comes from a separate image set (`oe_u8`), every sample labelled anomalous (`datasets/__init__.py:300`).
    The images are uploaded once; a task is an index list over them (`ResidentImageSource(normal_index=...)`), so iterating 30
    classes x 2 seeds does not copy the set 60 times."""

    def __init__(self, train_u8, train_classes, test_u8, test_classes, oe_u8, classes, crop, device="cuda", **source_kw):
        dev = torch.device(device)
        self.train, self.test, self.oe = (t.to(dev).contiguous() for t in (train_u8, test_u8, oe_u8))
        self.train_classes = torch.as_tensor(train_classes, dtype=torch.int64).clone()
        self.test_classes = torch.as_tensor(test_classes, dtype=torch.int64).clone()
        self.classes = list(classes)
        self.crop, self.source_kw, self.device = crop, dict(source_kw), dev

    def no_classes(self) -> int:
        return len(self.classes)

    def source(self, normal_classes, seed: int = 0) -> ResidentImageSource:
        src = ResidentImageSource(self.train, self.oe, self.test, ad_targets(self.test_classes, normal_classes), self.crop,
                                  seed=seed, device=self.device, normal_index=normal_subset(self.train_classes, normal_classes),
                                  **self.source_kw)
        src.normal_classes = tuple(int(c) for c in normal_classes)
        return src
