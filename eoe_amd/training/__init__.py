"""objective name -> trainer class, as `src/eoe/training/__init__.py:8-11`.  'clip' takes the frozen text features from the
caller (the text tower is outside the hot path); the autoencoder trainer is not built."""
from .ad_trainer import ADTrainer, NanGradientsError      # noqa: F401
from .hsc import HSCTrainer
from .bce import BCETrainer
from .dsvdd import DSVDDTrainer
from .dsad import DSADTrainer
from .focal import FocalTrainer
from .clip import ADClipTrainer

TRAINER = {"hsc": HSCTrainer, "bce": BCETrainer, "dsvdd": DSVDDTrainer, "dsad": DSADTrainer, "focal": FocalTrainer,
           "clip": ADClipTrainer}
