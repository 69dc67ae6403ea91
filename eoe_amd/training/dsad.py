"""Deep SAD objective, `src/eoe/training/dsad.py:7-21`: same three hooks on fused HIP kernels."""
from .. import ops
from .ad_trainer import ADTrainer


class DSADTrainer(ADTrainer):
    """deep semi-supervised AD with outlier exposure"""

    def prepare_metric(self, cstr, loader, model, seed, **kwargs):
        return None                                                   # dsad.py:9-10

    def compute_anomaly_score(self, features, center, train=False, **kwargs):
        return ops.hsc_score(features)                                # dsad.py:12-15: the HSC score

    def loss(self, features, labels, center, **kwargs):
        return ops.dsad_loss(features, labels, kwargs.get("nominal_label", 0), kwargs.get("inv_count", None))   # dsad.py:17-21
