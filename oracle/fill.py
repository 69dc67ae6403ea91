"""Deterministic, torch-RNG-independent tensor fill (test infrastructure; see oracle/__init__.py).

The reference never seeds and torch RNG streams are not portable across versions, so every
fixture weight / synthetic batch is produced by this counter-based rule instead: element i of the tensor
named ``name`` is  splitmix64(crc32(name) * 2^32 + i)  mapped to a uniform in [-1, 1) and scaled so that its
standard deviation equals ``std``.  The GPU box regenerates bit-identical tensors from (name, shape, std).
"""
import zlib
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def uniform_pm1(name: str, n: int) -> np.ndarray:
    """n float64 values uniform in [-1, 1), a pure function of (name, index)."""
    base = np.uint64(zlib.crc32(name.encode("utf-8"))) << np.uint64(32)
    idx = np.arange(n, dtype=np.uint64) + base
    bits = _splitmix64(idx) >> np.uint64(11)  # 53 random bits
    return bits.astype(np.float64) * (2.0 / float(1 << 53)) - 1.0


def fill(name: str, shape, std: float = 1.0, mean: float = 0.0) -> np.ndarray:
    """float32 array of `shape`, zero-mean uniform with standard deviation `std`, plus `mean`."""
    n = int(np.prod(shape)) if len(shape) else 1
    u = uniform_pm1(name, n) * (np.sqrt(3.0) * std) + mean
    return u.astype(np.float32).reshape(shape)


def fill_int(name: str, shape, lo: int, hi: int) -> np.ndarray:
    """int64 array with values in [lo, hi)."""
    n = int(np.prod(shape)) if len(shape) else 1
    base = np.uint64(zlib.crc32(name.encode("utf-8"))) << np.uint64(32)
    bits = _splitmix64(np.arange(n, dtype=np.uint64) + base) >> np.uint64(16)
    return (bits % np.uint64(hi - lo)).astype(np.int64).reshape(shape) + lo
