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
