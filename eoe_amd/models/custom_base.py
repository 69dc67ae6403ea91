"""`CustomNet`: an encoder wrapped with an optional prediction head, the reference's hook for user-defined models
(`src/eoe/models/custom_base.py:6-51`; discovered by `train_only_custom.py:23-26`).

Interface kept so that reference code and snapshots interchange: constructor `(feature_model_output_dim, prediction_head,
clf, freeze)`, attributes `feature_model`, `feature_dim`, `clf`, `prediction_head`, `freeze`, the head parameters
`final_linear.{weight,bias}` (-> 256 features for distance objectives such as HSC, -> 1 logit for classification objectives
such as BCE / focal), `freeze_parts()` and `load_feature_model_weights()`.  The head runs through this package's linear op.
"""
from abc import ABC, abstractmethod

import torch.nn as nn

from .. import ops

HEAD_WIDTH = 256          # feature width every distance-type objective trains on (custom_base.py:26)


class CustomNet(nn.Module, ABC):
    @abstractmethod
    def __init__(self, feature_model_output_dim: int, prediction_head: bool = True, clf: bool = False,
                 freeze: bool = False):
        """Subclasses call this first and then assign their encoder to `self.feature_model`."""
        super().__init__()
        self.feature_dim, self.prediction_head, self.clf, self.freeze = feature_model_output_dim, prediction_head, clf, freeze
        self.feature_model: nn.Module = nn.Identity()
        if prediction_head:
            # nn.Linear only as the parameter container (names and default init as in the reference)
            self.final_linear = nn.Linear(feature_model_output_dim, 1 if clf else HEAD_WIDTH)
        elif clf and feature_model_output_dim != 1:
            raise ValueError(f"{type(self).__name__}: a classification objective (BCE, focal) needs a single output neuron, but "
                             f"there is no prediction head and the feature model emits {feature_model_output_dim} values.")

    def freeze_parts(self) -> bool:
        """stop gradients at the encoder when constructed with freeze=True (custom_base.py:35-40); tells whether it did"""
        if not self.freeze:
            return False
        self.feature_model.requires_grad_(False)
        return True

    def load_feature_model_weights(self, model_state_dict: dict):
        self.feature_model.load_state_dict(model_state_dict)

    def forward(self, x):
        feats = self.feature_model(x)
        if not self.prediction_head:
            return feats
        return ops.linear(feats.flatten(1), self.final_linear.weight, self.final_linear.bias)
