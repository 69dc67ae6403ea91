"""CustomNet subclasses, discoverable the way `src/eoe/main/train_only_custom.py:23-26` discovers them
(`inspect.getmembers` on a `models.custom` module; ctor `(prediction_head, clf, freeze)` as `custom.py:6-8`).

The reference ships no CLIP CustomNet; "CLIP ViT-B/32 frozen encoder + HSC head" and "full fine-tune, BCE"
(BASELINE.json configs 4/5) are only expressible through this API."""
from .custom_base import CustomNet
from .clip_vit import VisualTransformer
from .resnet import WideResNet


class ClipViTB32Custom(CustomNet):
    """feature_model = CLIP ViT-B/32 image tower (224, patch 32, width 768, 12 layers, 12 heads, 512-d)"""

    def __init__(self, prediction_head: bool = True, clf: bool = False, freeze: bool = False, layers: int = 12,
                 input_resolution: int = 224):
        super().__init__(512, prediction_head, clf, freeze)
        self.feature_model = VisualTransformer(input_resolution, 32, 768, layers, 12, 512)


class WideResNetCustom(CustomNet):
    """the reference's own example CustomNet (`custom.py:5-8`): feature_model = WideResNet(256, False)"""

    def __init__(self, prediction_head: bool = True, clf: bool = False, freeze: bool = False):
        super().__init__(256, prediction_head, clf, freeze)
        self.feature_model = WideResNet(256, False)
