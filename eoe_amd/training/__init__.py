"""objective name -> trainer class, as `src/eoe/training/__init__.py:8-11` (objectives outside the north_star --
dsvdd, dsad, focal, ae, clip -- are not built)"""
from .ad_trainer import ADTrainer, NanGradientsError      # noqa: F401
from .hsc import HSCTrainer
from .bce import BCETrainer

TRAINER = {"hsc": HSCTrainer, "bce": BCETrainer}
