"""objective name -> trainer class, as `src/eoe/training/__init__.py:8-11`.  'clip' (the text-prompt objective, SURVEY.md
section 8f N2) and the autoencoder trainer are not built."""
from .ad_trainer import ADTrainer, NanGradientsError      # noqa: F401
from .hsc import HSCTrainer
from .bce import BCETrainer
from .dsvdd import DSVDDTrainer
from .dsad import DSADTrainer
from .focal import FocalTrainer

TRAINER = {"hsc": HSCTrainer, "bce": BCETrainer, "dsvdd": DSVDDTrainer, "dsad": DSADTrainer, "focal": FocalTrainer}
