"""HSC objective, `src/eoe/training/hsc.py:7-21`: same three hooks, each one fused HIP kernel (eoe_hsc_*)."""
import torch

from .. import ops
from .ad_trainer import ADTrainer


class HSCTrainer(ADTrainer):
    """hypersphere classifier for semi-supervised AD with outlier exposure"""

    def prepare_metric(self, cstr, loader, model, seed, **kwargs):
        return None                                                   # hsc.py:9-10

    def compute_anomaly_score(self, features, center, train=False, **kwargs):
        return ops.hsc_score(features)                                # hsc.py:12-15

    def loss(self, features, labels, center, **kwargs):
        # hsc.py:17-21; `inv_count` (1/global batch) is the data-parallel extension, default = plain mean
        return ops.hsc_loss(features, labels, kwargs.get("nominal_label", 0), kwargs.get("inv_count", None))
