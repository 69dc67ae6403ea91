"""Deep SVDD objective, `src/eoe/training/dsvdd.py:7-27`."""
import torch

from .. import ops
from .ad_trainer import ADTrainer


class DSVDDTrainer(ADTrainer):
    """deep support vector data description (unsupervised)"""

    def prepare_metric(self, cstr, loader, model, seed, **kwargs):
        """the centre (dsvdd.py:10-22): mean over the batches of the per-batch mean feature of the samples labelled 0, entries
        closer to zero than eps pushed to +-eps.  One forward pass over the loader before training -- like the reference's,
        in the model's current mode and on the loader's images as they come: the device-side Normalize of the training loop
        (`ad_trainer.py:413-425`) is not part of that pass, so a fused normalise is switched off for it."""
        eps = kwargs.get("eps", 1e-1)
        means = []
        enc = getattr(model, "feature_model", model)
        if hasattr(enc, "set_normalize"):
            enc.set_normalize(None, None)
        was_training = model.training
        for batch in loader:
            imgs, lbls = batch[0].to(self.device), batch[1]
            with torch.no_grad():
                feats = model(imgs[(lbls == 0).to(imgs.device)])
            means.append(feats.float().mean(0, keepdim=True))
        model.train(was_training)
        center = torch.cat(means).mean(0, keepdim=True)
        center[(center.abs() < eps) & (center < 0)] = -eps
        center[(center.abs() < eps) & (center > 0)] = eps
        return center.to(self.device)

    def compute_anomaly_score(self, features, center, train=False, **kwargs):
        return ops.dsvdd_score(features, center)                      # dsvdd.py:24-25

    def loss(self, features, labels, center, **kwargs):
        return ops.dsvdd_loss(features, center, kwargs.get("inv_count", None))      # dsvdd.py:26-27
