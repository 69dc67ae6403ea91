"""BCE objective, `src/eoe/training/bce.py:9-20`."""
from .. import ops
from .ad_trainer import ADTrainer


class BCETrainer(ADTrainer):
    """binary cross entropy with logits for semi-supervised AD with outlier exposure"""

    def prepare_metric(self, cstr, loader, model, seed, **kwargs):
        return None                                                   # bce.py:12-13

    def compute_anomaly_score(self, features, center, train=False, **kwargs):
        return ops.bce_score(features, kwargs.get("nominal_label", 0))   # bce.py:15-17

    def loss(self, features, labels, center, **kwargs):
        return ops.bce_loss(features, labels, kwargs.get("inv_count", None))   # bce.py:19-20
