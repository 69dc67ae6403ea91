"""CPU restatement of the step-batch contract (test infrastructure; integer paths are bit-exact).

Follows reference `src/eoe/datasets/bases.py:570-600` (BalancedConcatLoader): a step batch is
`cat([normal_i, oe_i[:len(normal_i)]])` for images, labels and indices; OE indices are offset by the length
of the normal dataset (`bases.py:596`); if the OE set is smaller than the normal set its index list is tiled
`ceil(len(normal)/len(oe))` times (`bases.py:580-584`); the loader's length is the normal loader's
(`bases.py:599-600`).  And `src/eoe/training/ad_trainer.py:413-425` + `utils/transformations.py:126-138`: every
image gets one per-channel affine `(x - mean_c) / std_c` with the normal class's statistics (SURVEY.md
section 8a note 1).
"""
import numpy as np


def tile_oe_indices(oe_indices: np.ndarray, n_normal: int) -> np.ndarray:
    oe_indices = np.asarray(oe_indices, dtype=np.int64)
    if len(oe_indices) < n_normal:
        r = int(np.ceil(n_normal / len(oe_indices)))
        oe_indices = np.tile(oe_indices.reshape(1, -1), (r, 1)).reshape(-1)
    return oe_indices


def balanced_concat(normal, oe_chunks, n_normal_dataset: int):
    """normal = (imgs, lbls, idcs) of one normal batch; oe_chunks = iterator of (imgs, lbls, idcs) OE batches.
    Returns the concatenated (imgs, lbls, idcs) exactly as BalancedConcatLoader.__next__ does."""
    oe = [np.asarray(a) for a in next(oe_chunks)]
    while oe[1].shape[0] < normal[1].shape[0]:
        nxt = next(oe_chunks)
        oe = [np.concatenate([a, np.asarray(b)]) for a, b in zip(oe, nxt)]
    oe[-1] = oe[-1] + n_normal_dataset
    n = normal[0].shape[0]
    return [np.concatenate([np.asarray(i), j[:n]]) for i, j in zip(normal, oe)]


def synthetic_labels(n_normal: int, n_oe: int) -> np.ndarray:
    """labels of a synthetic step batch: nominal 0 for the normal half, 1 for the OE half."""
    return np.concatenate([np.zeros(n_normal, np.int64), np.ones(n_oe, np.int64)])


def normalize(imgs: np.ndarray, mean, std) -> np.ndarray:
    """per-channel affine on an N x C x H x W float32 batch (transformations.py:126-138)."""
    mean = np.asarray(mean, np.float32).reshape(1, -1, 1, 1)
    std = np.asarray(std, np.float32).reshape(1, -1, 1, 1)
    return (imgs.astype(np.float32) - mean) / std


def shard_rows(n_normal: int, n_oe: int, rank: int, world: int):
    """data-parallel partition of one step batch: rank r takes rows
    [r*B/R, (r+1)*B/R) of the normal half and of the OE half; returns the global row indices."""
    def part(n):
        lo = (n * rank) // world
        hi = (n * (rank + 1)) // world
        return lo, hi
    a, b = part(n_normal)
    c, d = part(n_oe)
    return np.concatenate([np.arange(a, b), n_normal + np.arange(c, d)]).astype(np.int64)


# ---------------------------------------------------------------------------------------------------------------------
# the class x seed loop's task definition (reference `training/ad_trainer.py:166-175`, `datasets/bases.py:129-139,169-203`)
# ---------------------------------------------------------------------------------------------------------------------
def nominal_classes(ad_mode: str, cur_class: int, n_classes: int):
    """ADTrainer.get_nominal_classes (ad_trainer.py:166-175): the normal classes of the task "class cur_class" """
    if ad_mode == "one_vs_rest":
        return [cur_class]
    if ad_mode == "leave_one_out":
        return [c for c in range(n_classes) if c != cur_class]
    if ad_mode == "fifty_fifty":
        return [c % n_classes for c in range(cur_class, n_classes // 2 + cur_class)]
    raise NotImplementedError(ad_mode)


def normal_subset(class_labels, normal_classes) -> np.ndarray:
    """TorchvisionDataset.create_subset without a sample limit (bases.py:192-203): ascending rows whose class is normal"""
    return np.argwhere(np.isin(np.asarray(class_labels), list(normal_classes))).flatten().astype(np.int64)


def ad_targets(class_labels, normal_classes, nominal_label: int = 0) -> np.ndarray:
    """the target_transform of bases.py:129-139: anomalous iff the class is not one of the normal classes"""
    normal = np.isin(np.asarray(class_labels), list(normal_classes))
    return np.where(normal, nominal_label, 1 - nominal_label).astype(np.int64)
