"""Drop-in for `src/eoe/models/custom_base.py:6-51` (CustomNet): feature_model + optional final_linear
(-> 256 for HSC-like objectives, -> 1 for classification objectives) + freeze_parts().  Same constructor,
attribute and parameter names (`feature_model.*`, `final_linear.{weight,bias}`)."""
from abc import ABC, abstractmethod

import torch.nn as nn

from .. import ops


class CustomNet(nn.Module, ABC):
    @abstractmethod
    def __init__(self, feature_model_output_dim: int, prediction_head: bool = True, clf: bool = False,
                 freeze: bool = False):
        super().__init__()
        self.feature_model: nn.Module = nn.Identity()  # implement this
        self.feature_dim = feature_model_output_dim
        self.clf = clf
        self.prediction_head = prediction_head
        self.freeze = freeze
        if self.prediction_head:
            self.final_linear = nn.Linear(self.feature_dim, 1 if self.clf else 256)   # parameter container
        elif self.clf and feature_model_output_dim != 1:
            raise ValueError(
                f"{self.__class__} was created for a classification loss (BCE, focal, ...) without an additional "
                f"prediction head while its feature model predicts more than one neuron ({self.feature_dim} > 1)."
            )

    def freeze_parts(self) -> bool:
        # custom_base.py:35-40
        if self.freeze:
            for n, p in self.feature_model.named_parameters():
                p.requires_grad_(False)
            return True
        return False

    def load_feature_model_weights(self, model_state_dict: dict):
        self.feature_model.load_state_dict(model_state_dict)

    def forward(self, x):
        # custom_base.py:45-51
        features = self.feature_model(x)
        if self.prediction_head:
            out = ops.linear(features.flatten(1), self.final_linear.weight, self.final_linear.bias)
        else:
            out = features
        return out
