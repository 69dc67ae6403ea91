"""eoe_amd -- MI355X-native (gfx950) implementation of the outlier-exposure AD training hot path of
liznerski/eoe: the ADTrainer inner loop (mixed normal+OE batch -> encoder -> anomaly score -> HSC / BCE loss ->
backward -> Adam) as hand-written HIP kernels behind a C ABI (include/eoe_hip.h, libeoe_hip.so), surfaced through
torch.autograd.Function bindings so the reference's nn.Module / trainer-hook / optimizer API stays unchanged.

There is no CPU fallback: importing this package loads (or builds) the HIP library and fails loudly otherwise.
"""
import os as _os
import sys as _sys

# Kernel arguments in device memory (HIP runtime setting, read when libamdhip64 is loaded -- at `import torch`): worth 4 % of the ViT step
# (bench.py has the measurement).  Only effective when this package is imported BEFORE torch; a process that imported torch first should
# export HIP_FORCE_DEV_KERNARG=1 itself (bench.py, the tests and __graft_entry__ do).
if "torch" not in _sys.modules:
    _os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

from . import _lib                      # noqa: F401,E402  (loads libeoe_hip.so; raises if unavailable)
from .ops import set_compute_dtype, compute_dtype, hsc_loss, hsc_score, bce_loss, bce_score, linear  # noqa: F401
from .ops import dsad_loss, dsvdd_loss, dsvdd_score, focal_loss, set_parity_mode, parity_mode  # noqa: F401
from .ops import set_grad_scale, grad_scale, default_grad_scale  # noqa: F401
from .optim import FusedAdam, FusedSGD  # noqa: F401
from .graph import GraphedStep          # noqa: F401

__all__ = ["set_compute_dtype", "compute_dtype", "hsc_loss", "hsc_score", "bce_loss", "bce_score", "linear",
           "FusedAdam", "FusedSGD", "GraphedStep", "set_parity_mode", "parity_mode", "set_grad_scale", "grad_scale",
           "default_grad_scale"]
