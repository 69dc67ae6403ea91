"""CPU restatement of the ADTrainer inner loop (test infrastructure; see oracle/__init__.py).

Follows reference `src/eoe/training/ad_trainer.py:406-455` (train_cls: per batch  opt.zero_grad -> features =
model(imgs) -> loss -> backward -> opt.step -> scores from the *pre-step* features; per epoch  NaN check,
ROC-AUC on the concatenated labels/scores, sched.step) and `:498-522` (eval_cls scoring).  The optimiser is
this package's own Adam restatement (oracle/optim.py), the objectives are oracle/objectives.py.
"""
from typing import Callable, List, Sequence
import numpy as np
import torch

from . import objectives, optim, metrics

OBJECTIVES = {
    "hsc": (lambda f, y: objectives.hsc_loss(f, y, 0), lambda f: objectives.hsc_score(f)),
    "bce": (lambda f, y: objectives.bce_loss(f, y), lambda f: objectives.bce_score(f, 0)),
}


def train_steps(model: torch.nn.Module, batches: Sequence, objective: str = "hsc", lr: float = 1e-3,
                weight_decay: float = 0.0, milestones: Sequence[int] = (), steps_per_epoch: int = None,
                collect_grads: bool = False):
    """run one optimisation step per (imgs, labels) in `batches`; returns dict with per-step 'loss',
    'scores' (from pre-step features), per-epoch 'auc', and optionally the first step's gradients."""
    loss_fn, score_fn = OBJECTIVES[objective]
    model.train()
    if hasattr(model, "freeze_parts"):
        model.freeze_parts()                                   # ad_trainer.py:593-596 (via self.load)
    params = [p for p in model.parameters()]
    state = optim.AdamState(params)
    out = {"loss": [], "scores": [], "auc": [], "grads": None}
    spe = steps_per_epoch or len(batches)
    ep_labels, ep_scores = [], []
    for it, (imgs, lbls) in enumerate(batches):
        epoch = it // spe
        cur_lr = optim.multistep_lr(lr, list(milestones), epoch)
        for p in params:
            p.grad = None                                      # opt.zero_grad()  (:428)
        feats = model(imgs)                                    # :429
        loss = loss_fn(feats, lbls)                            # :430
        loss.backward()                                        # :431
        if collect_grads and out["grads"] is None:
            out["grads"] = {n: (p.grad.detach().clone() if p.grad is not None else None)
                            for n, p in model.named_parameters()}
        optim.adam_step(params, [p.grad for p in params], state, cur_lr, weight_decay)   # :432
        with torch.no_grad():
            scores = score_fn(feats.detach())                  # :434-436 (pre-step features)
        out["loss"].append(float(loss.item()))
        out["scores"].append(scores.numpy().copy())
        ep_labels.append(lbls.numpy())
        ep_scores.append(scores.numpy())
        if (it + 1) % spe == 0 or it == len(batches) - 1:
            la, sc = np.concatenate(ep_labels), np.concatenate(ep_scores)
            if np.isnan(sc).any():
                raise RuntimeError("NaN scores")              # NanGradientsError (:448-449)
            out["auc"].append(metrics.roc_auc(la, sc) if (la == 1).any() else float("nan"))   # :452-454
            ep_labels, ep_scores = [], []
    return out


def eval_scores(model: torch.nn.Module, batches: Sequence, objective: str = "hsc"):
    """eval_cls scoring (:498-522): no_grad forward in eval mode, scores, ROC-AUC + average precision."""
    _, score_fn = OBJECTIVES[objective]
    model.eval()
    la, sc = [], []
    with torch.no_grad():
        for imgs, lbls in batches:
            sc.append(score_fn(model(imgs)).numpy())
            la.append(lbls.numpy())
    la, sc = np.concatenate(la), np.concatenate(sc)
    return {"scores": sc, "labels": la, "auc": metrics.roc_auc(la, sc),
            "avg_prec": metrics.average_precision(la, sc)}


def synthetic_batch(tag: str, n_normal: int, n_oe: int, res: int, shift: float = 0.5):
    """already-normalised synthetic step batch: zero-mean unit-std values from
    oracle.fill; the OE half gets a fixed +shift * pattern so that AUC is meaningful."""
    from . import fill, batching
    n = n_normal + n_oe
    x = fill.fill(f"{tag}/imgs", (n, 3, res, res), std=1.0)
    pat = fill.fill("pattern/oe", (1, 3, res, res), std=1.0)
    x[n_normal:] += shift * pat
    y = batching.synthetic_labels(n_normal, n_oe)
    return torch.from_numpy(x), torch.from_numpy(y)
