"""Focal-loss objective, `src/eoe/training/focal.py:11-36`: encoder with the 1-wide clf head."""
from .. import ops
from .ad_trainer import ADTrainer


class FocalTrainer(ADTrainer):
    """focal loss (gamma 2, eps 1e-7) for semi-supervised AD with outlier exposure"""

    def prepare_metric(self, cstr, loader, model, seed, **kwargs):
        return None                                                   # focal.py:27-28

    def compute_anomaly_score(self, features, center, train=False, **kwargs):
        return ops.bce_score(features, kwargs.get("nominal_label", 0))            # focal.py:30-32

    def loss(self, features, labels, center, **kwargs):
        return ops.focal_loss(features, labels, kwargs.get("inv_count", None))      # focal.py:34-36
