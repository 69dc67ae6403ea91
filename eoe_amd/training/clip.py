"""CLIP text-prompt objective, `src/eoe/training/clip.py:13-103` on fused HIP kernels.

The reference's `prepare_metric` (clip.py:50-64) tokenises two (one_vs_rest) or C (leave_one_out) prompts and runs them through
CLIP's *text* tower once; the text tower, the tokenizer and the checkpoint download are outside the hot path (SURVEY.md
section 2: I/O, network), so this trainer takes the frozen text features from the caller: `text_features` is either a tensor
[T, d] or a callable `(cstr, ad_mode) -> tensor`; they are l2-normalised here exactly as clip.py:62.  Everything the training loop
does with them -- the loss, the anomaly score, SGD with Nesterov momentum for the image tower (ad_trainer.py:380-381) -- runs on
the GPU kernels (`eoe_clip_fwd/bwd/score`, `eoe_sgd_multi`)."""
import torch

from .. import ops
from ..optim import FusedSGD
from .ad_trainer import ADTrainer


class ADClipTrainer(ADTrainer):
    def __init__(self, model, *args, text_features=None, fp16_weights=False, **kwargs):
        """fp16_weights: the reference's CLIP towers carry fp16 convolution / linear / attention / projection parameters on a GPU
        (`convert_weights`, clip/model.py:371-392, applied by build_model :430; `clip.load` undoes it on the CPU only) and SGD updates those
        fp16 tensors; True reproduces that arithmetic (`eoe_amd.models.convert_weights` on the image tower + `eoe_sgd_multi`'s fp16 path).
        The default keeps fp32 masters, which is what the reference gets on a CPU and is strictly more accurate."""
        super().__init__(model, *args, **kwargs)
        self.text_features = text_features
        self.fp16_weights = bool(fp16_weights)

    def make_optimizer(self, model):
        # ad_trainer.py:380-381: CLIP models are trained with SGD(momentum 0.9, nesterov)
        if self.fp16_weights:
            from ..models import convert_weights
            convert_weights(getattr(model, "feature_model", model))          # the CLIP tower; a CustomNet head is created in fp32
        return FusedSGD(model.parameters(), lr=self.lr, weight_decay=self.wdk, momentum=0.9, nesterov=True)

    def prepare_metric(self, cstr, loader, model, seed, **kwargs):
        t = self.text_features(cstr, self.ad_mode) if callable(self.text_features) else self.text_features
        if t is None:
            raise RuntimeError("ADClipTrainer needs the frozen text features of the prompts (clip.py:50-64): pass text_features=")
        t = torch.as_tensor(t, dtype=torch.float32).to(self.device)
        expect = 2 if self.ad_mode == "one_vs_rest" else None
        if t.dim() != 2 or (expect is not None and t.shape[0] != expect):
            raise ValueError(f"text_features must be [T, d] (T = 2 for one_vs_rest), got {tuple(t.shape)}")
        return t / t.norm(dim=-1, keepdim=True)                              # clip.py:62

    def compute_anomaly_score(self, features, center, train=False, **kwargs):
        return ops.clip_score(features, center)                               # clip.py:66-79

    def loss(self, features, labels, center, **kwargs):
        return ops.clip_loss(features, labels, center, kwargs.get("nominal_label", 0), self.ad_mode == "leave_one_out",
                             kwargs.get("inv_count", None))                   # clip.py:81-103
