from .clip_vit import VisualTransformer, ResidualAttentionBlock, convert_weights      # noqa: F401
from .custom_base import CustomNet                                   # noqa: F401
from .custom import ClipViTB32Custom                                 # noqa: F401
from .cnn import CNN32, CNN28                                       # noqa: F401
from .resnet import WideResNet, BasicBlock                          # noqa: F401
from .cbam import CBAM, ChannelGate, SpatialGate                    # noqa: F401
