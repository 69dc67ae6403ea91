#!/usr/bin/env python
"""Generates the golden fixtures under tests/golden/ by running the REFERENCE's own modules.

Run in the build container only (needs /root/reference; never on the GPU box):
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What is reference code here: `eoe.models.cnn.CNN32`, `eoe.models.custom_base.CustomNet`, and
`clip/model.py`'s `VisualTransformer` / `ResidualAttentionBlock` (imported from /root/reference/src), driven by
the stock third-party calls the reference's trainer makes (`torch.optim.Adam`, `MultiStepLR`,
`binary_cross_entropy_with_logits`, `torch.norm`, `sklearn.metrics.roc_curve/auc/average_precision_score`;
`src/eoe/training/ad_trainer.py:8,383-384,453-454,517-521`, `hsc.py:13-21`, `bce.py:16-20`).
`eoe.training.*` and `eoe.datasets.*` cannot be imported here (ordinary ModuleNotFoundError: torchvision,
kornia, cv2, tensorboard -- SURVEY.md section 8c), so the trainer loop / objectives are driven through those
stock calls in this script.  Weights and inputs come from oracle.fill (a pure function of name/shape), so the
fixtures hold only small outputs.  Only data is written; no reference source is copied.
"""
import importlib.util
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/src")
sys.dont_write_bytecode = True

from oracle import fill, models as omodels, trainer as otrainer  # noqa: E402  (only for the fill rule / batches)

from eoe.models.cnn import CNN32 as RefCNN32, CNN28 as RefCNN28    # noqa: E402
from eoe.models.custom_base import CustomNet as RefCustomNet       # noqa: E402

_spec = importlib.util.spec_from_file_location(
    "ref_clip_model", "/root/reference/src/eoe/models/clip_official/clip/model.py")
ref_clip = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(ref_clip)

torch.set_num_threads(8)
torch.manual_seed(0)


def hsc_loss_ref(f, y):
    d = torch.sqrt(torch.norm(f, p=2, dim=1) ** 2 + 1) - 1
    s = 1 - torch.exp(-d)
    return torch.where(y == 0, d, -torch.log(s + 1e-9)).mean()


def hsc_score_ref(f):
    d = torch.sqrt(torch.norm(f, p=2, dim=1) ** 2 + 1) - 1
    return 1 - torch.exp(-d)


def bce_loss_ref(f, y):
    return F.binary_cross_entropy_with_logits(f.squeeze(), y.float())


def bce_score_ref(f):
    return torch.sigmoid(f).squeeze()


LOSS = {"hsc": (hsc_loss_ref, hsc_score_ref), "bce": (bce_loss_ref, bce_score_ref)}


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


def grad_summary(model):
    """per-tensor gradient norm, sum and first 16 values (full tensors would be too large to commit)"""
    out = {}
    for n, p in model.named_parameters():
        if p.grad is None:
            continue
        g = p.grad.detach().double()
        out[f"gnorm/{n}"] = g.norm().item()
        out[f"gsum/{n}"] = g.sum().item()
        out[f"ghead/{n}"] = p.grad.detach().flatten()[:16].numpy().copy()
    return out


def run_trajectory(model, batches, objective, lr, wd, freeze=False):
    """the reference inner loop (ad_trainer.py:428-436) with the stock Adam"""
    loss_fn, score_fn = LOSS[objective]
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=lr, weight_decay=wd, amsgrad=False)
    if freeze:
        model.freeze_parts()
    losses, scores, first = [], [], None
    for imgs, lbls in batches:
        opt.zero_grad()
        feats = model(imgs)
        loss = loss_fn(feats, lbls)
        loss.backward()
        if first is None:
            first = grad_summary(model)
            first["features0"] = feats.detach().numpy().copy()
            for n, b in model.named_buffers():      # BN running stats after the first forward
                first[f"buf0/{n}"] = b.numpy().copy()
        opt.step()
        opt.zero_grad()
        losses.append(loss.item())
        scores.append(score_fn(feats.detach()).numpy().copy())
    return np.array(losses, np.float64), np.stack(scores), first


# ----------------------------------------------------------------------------------------------- G1 objectives
def g1():
    f = torch.from_numpy(fill.fill("g1/features", (16, 256), std=0.08))
    y = torch.from_numpy(fill.fill_int("g1/labels", (16,), 0, 2))
    out = {}
    for name in ("hsc", "bce"):
        ff = (f if name == "hsc" else f[:, :1] * 20).clone().requires_grad_(True)
        loss = LOSS[name][0](ff, y)
        loss.backward()
        out[f"{name}_loss"] = loss.item()
        out[f"{name}_scores"] = LOSS[name][1](ff.detach()).numpy()
        out[f"{name}_grad"] = ff.grad.numpy()
    # known answers
    z = torch.zeros(2, 256)
    out["hsc_zero_nominal"] = hsc_loss_ref(z, torch.zeros(2, dtype=torch.long)).item()
    out["hsc_zero_oe"] = hsc_loss_ref(z, torch.ones(2, dtype=torch.long)).item()
    r3 = torch.zeros(1, 256)
    r3[0, :3] = 1.0
    out["hsc_sqrt3_score"] = hsc_score_ref(r3).numpy()
    save("g1_objectives", **out)


# ----------------------------------------------------------------------------------------------- G9 other objectives (N4)
def dsad_loss_ref(f, y, nominal=0):                     # training/dsad.py:17-21
    dists = torch.norm(f, p=2, dim=1) ** 2
    return torch.where(y == nominal, dists, ((dists + 1e-9) ** (-1))).mean()


def dsvdd_loss_ref(f, c):                               # training/dsvdd.py:24-27
    return (f - c).pow(2).sum(-1).mean()


def dsvdd_center_ref(batch_feats, eps=1e-1):            # training/dsvdd.py:10-22 on already-computed nominal features
    center = torch.cat([bf.mean(0).unsqueeze(0) for bf in batch_feats]).mean(0).unsqueeze(0)
    center[(abs(center) < eps) & (center < 0)] = -eps
    center[(abs(center) < eps) & (center > 0)] = eps
    return center


def focal_loss_ref(x, y, gamma=2.0, eps=1e-7):          # training/focal.py:11-24,34-36
    bce = torch.nn.functional.binary_cross_entropy_with_logits(x.squeeze(), y.float(), reduction='none')
    pt = torch.exp(-bce).clamp(eps, 1. - eps)
    return ((1 - pt).pow(gamma) * bce).mean()


def g9():
    f = torch.from_numpy(fill.fill("g9/features", (16, 256), std=0.08))
    y = torch.from_numpy(fill.fill_int("g9/labels", (16,), 0, 2))
    out = {}
    ff = f.clone().requires_grad_(True)
    loss = dsad_loss_ref(ff, y)
    loss.backward()
    out["dsad_loss"], out["dsad_grad"] = loss.item(), ff.grad.numpy()
    out["dsad_scores"] = hsc_score_ref(f).numpy()                                   # dsad.py:12-15 = the HSC score
    feats = [torch.from_numpy(fill.fill(f"g9/cb{i}", (5 + i, 256), std=0.3, mean=0.02)) for i in range(3)]
    c = dsvdd_center_ref(feats)
    out["dsvdd_center"] = c.numpy()
    ff = f.clone().requires_grad_(True)
    loss = dsvdd_loss_ref(ff, c)
    loss.backward()
    out["dsvdd_loss"], out["dsvdd_grad"] = loss.item(), ff.grad.numpy()
    out["dsvdd_scores"] = (f - c).pow(2).sum(-1).numpy()
    x = torch.cat([f[:, :1] * 20, torch.tensor([[40.0], [-40.0], [0.0], [18.0]])])     # incl. saturated logits (pt clamp)
    yy = torch.cat([y, torch.tensor([1, 0, 1, 0])])
    xx = x.clone().requires_grad_(True)
    loss = focal_loss_ref(xx, yy)
    loss.backward()
    out["focal_x"], out["focal_y"] = x.numpy(), yy.numpy()
    out["focal_loss"], out["focal_grad"] = loss.item(), xx.grad.numpy()
    out["focal_scores"] = torch.sigmoid(x).squeeze().numpy()
    save("g9_objectives", **out)


# ----------------------------------------------------------------------------------------------- G2 CNN32
def g2():
    for clf, obj in ((False, "hsc"), (True, "bce")):
        m = RefCNN32(bias=True, clf=clf)
        omodels.deterministic_init(m, tag="cnn32")
        batches = [otrainer.synthetic_batch(f"g2/b{i}", 8, 8, 32) for i in range(5)]
        losses, scores, first = run_trajectory(m, batches, obj, lr=1e-3, wd=0.0)
        save(f"g2_cnn32_{obj}", losses=losses, scores=scores, **first)


# ----------------------------------------------------------------------------------------------- G3 ViT
class RefClipNet(RefCustomNet):
    def __init__(self, layers, clf=False, freeze=False):
        super().__init__(512, prediction_head=True, clf=clf, freeze=freeze)
        self.feature_model = ref_clip.VisualTransformer(224, 32, 768, layers, 12, 512)


def g3():
    for layers, n_half, steps, obj, freeze in ((2, 2, 3, "hsc", False), (2, 2, 3, "bce", False),
                                               (2, 2, 3, "hsc", True), (12, 1, 2, "hsc", False)):
        m = RefClipNet(layers, clf=(obj == "bce"), freeze=freeze)
        omodels.deterministic_init(m, tag="vit", layers=layers)
        batches = [otrainer.synthetic_batch(f"g3/b{i}", n_half, n_half, 224) for i in range(steps)]
        # lr / wd of the CLIP runner defaults (train_clip_imagenet.py:13-17)
        losses, scores, first = run_trajectory(m, batches, obj, lr=1e-4, wd=1e-3, freeze=freeze)
        tag = f"g3_vit_l{layers}_{obj}" + ("_frozen" if freeze else "")
        with torch.no_grad():
            m.eval()
            enc = m.feature_model(batches[0][0]).numpy()       # encoder output after the K steps
        save(tag, losses=losses, scores=scores, enc_after=enc, **first)


# ----------------------------------------------------------------------------------------------- G4 block
def g4():
    blk = ref_clip.ResidualAttentionBlock(768, 12)
    omodels.deterministic_init(blk, tag="blk", layers=12)
    x = torch.from_numpy(fill.fill("g4/x", (50, 2, 768), std=1.0)).requires_grad_(True)   # LND (model.py:227)
    y = blk(x)
    w = torch.from_numpy(fill.fill("g4/dy", (50, 2, 768), std=1.0))
    (y * w).sum().backward()
    save("g4_block", y=y.detach().numpy(), dx=x.grad.numpy(), **grad_summary(blk))


# ----------------------------------------------------------------------------------------------- G5 WideResNet
def g5():
    import types
    # resnet.py:3 imports torchvision.models.wide_resnet50_2 only for the out-of-scope WideResNet50Pretrained
    tv, tvm = types.ModuleType("torchvision"), types.ModuleType("torchvision.models")
    tvm.wide_resnet50_2 = None
    tv.models = tvm
    sys.modules.setdefault("torchvision", tv)
    sys.modules.setdefault("torchvision.models", tvm)
    from eoe.models.resnet import WideResNet as RefWRN
    from eoe.models.cbam import CBAM as RefCBAM
    m = RefWRN()
    omodels.deterministic_init(m, tag="wrn")
    batches = [otrainer.synthetic_batch(f"g5/b{i}", 2, 2, 224) for i in range(2)]
    losses, scores, first = run_trajectory(m, batches, "hsc", lr=1e-3, wd=0.0)      # train_imagenet.py:16-17
    save("g5_wideresnet_hsc", losses=losses, scores=scores, **first)
    # one CBAM(64) block on a 2x64x14x14 map, forward + all gradients
    cb = RefCBAM(64, 16)
    omodels.deterministic_init(cb, tag="cbam")
    cb.train()
    x = torch.from_numpy(fill.fill("g5/cbam_x", (2, 64, 14, 14), std=1.0)).requires_grad_(True)
    w = torch.from_numpy(fill.fill("g5/cbam_dy", (2, 64, 14, 14), std=1.0))
    y = cb(x)
    (y * w).sum().backward()
    extra = {f"buf/{n}": b.numpy().copy() for n, b in cb.named_buffers()}
    save("g5_cbam", y=y.detach().numpy(), dx=x.grad.numpy(), **grad_summary(cb), **extra)


# ----------------------------------------------------------------------------------------------- G7 metrics
def g7():
    from sklearn.metrics import roc_curve, auc, average_precision_score
    out = {}
    for i, (n, ties) in enumerate(((64, False), (200, True), (1000, True), (10, False))):
        s = fill.fill(f"g7/s{i}", (n,), std=1.0)
        if ties:
            s = np.round(s * 4) / 4
        y = fill.fill_int(f"g7/y{i}", (n,), 0, 2)
        y[0], y[1] = 0, 1
        fpr, tpr, _ = roc_curve(y, s)
        out[f"auc{i}"] = auc(fpr, tpr)
        out[f"ap{i}"] = average_precision_score(y, s)
        out[f"n{i}"] = n
        out[f"ties{i}"] = int(ties)
    save("g7_metrics", **out)


# ----------------------------------------------------------------------------------------------- G8 Adam / LR
def g8():
    out = {}
    for wd in (0.0, 1e-3):
        ps = [torch.nn.Parameter(torch.from_numpy(fill.fill(f"g8/p{i}", s, std=0.5)))
              for i, s in enumerate(((7, 5), (33,), (4, 3, 2)))]
        opt = torch.optim.Adam(ps, lr=1e-2, weight_decay=wd, amsgrad=False)
        for t in range(5):
            opt.zero_grad()
            for i, p in enumerate(ps):
                if i == 1 and t in (1, 2):
                    p.grad = None                      # a param without grad is skipped, its step does not advance
                else:
                    p.grad = torch.from_numpy(fill.fill(f"g8/g{i}/t{t}", tuple(p.shape), std=0.1))
            opt.step()
        for i, p in enumerate(ps):
            out[f"wd{wd}/p{i}"] = p.detach().numpy().copy()
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.Adam([p], lr=1e-3)
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, [3, 5, 6], 0.1)
    lrs = []
    for ep in range(9):
        lrs.append(sched.get_last_lr()[0])
        opt.step()
        sched.step()
    out["multistep_lrs"] = np.array(lrs)
    save("g8_adam", **out)


# ----------------------------------------------------------------------------------------------- G10 CLIP objective + SGD (N2)
def clip_loss_ref(image_features, labels, text_features, nominal_label=0, ad_mode="one_vs_rest"):   # training/clip.py:81-103
    anom_label = 1 - nominal_label
    image_features = image_features / image_features.norm(dim=-1, keepdim=True)
    similarity = (100.0 * image_features @ text_features.T).log_softmax(dim=-1)
    aloss = similarity[labels == anom_label][:, -1]
    if ad_mode == "one_vs_rest":
        nloss = similarity[labels == nominal_label][:, 0]
    else:
        nloss = similarity[labels == nominal_label][:, :-1].max(-1)[0]
    loss = torch.zeros_like(similarity[:, 0])
    loss[labels == anom_label] = aloss
    loss[labels == nominal_label] = nloss
    return loss.mul(-1).mean()


def clip_score_ref(image_features, center):                                                        # training/clip.py:66-79
    text_features = center / center.norm(dim=-1, keepdim=True)
    image_features = image_features / image_features.norm(dim=-1, keepdim=True)
    return (100.0 * image_features @ text_features.T).softmax(dim=-1)[:, -1]


def g10():
    out = {}
    f = torch.from_numpy(fill.fill("g10/features", (24, 512), std=0.4))
    y = torch.from_numpy(fill.fill_int("g10/labels", (24,), 0, 2))
    y[5] = 7                                            # a label that is neither nominal nor anomalous: loss 0 (clip.py:89-91)
    for mode, T in (("one_vs_rest", 2), ("leave_one_out", 30)):
        t = torch.from_numpy(fill.fill(f"g10/text{T}", (T, 512), std=1.0))
        t = t / t.norm(dim=-1, keepdim=True)            # prepare_metric, clip.py:62
        t = t * 0.25 + 0.75 * t[:1]                     # prompts of one dataset are close to each other: soft, non-saturated softmax
        t = t / t.norm(dim=-1, keepdim=True)
        for nominal in (0, 1):
            ff = f.clone().requires_grad_(True)
            loss = clip_loss_ref(ff, y, t, nominal, mode)
            loss.backward()
            out[f"{mode}/n{nominal}/loss"], out[f"{mode}/n{nominal}/grad"] = loss.item(), ff.grad.numpy().copy()
        out[f"{mode}/scores"] = clip_score_ref(f, t * 3.0).numpy()        # un-normalised centre: the score normalises it again
    # SGD with Nesterov momentum as constructed for CLIP models (ad_trainer.py:380-381), incl. a parameter without gradient
    for wd in (0.0, 1e-3):
        ps = [torch.nn.Parameter(torch.from_numpy(fill.fill(f"g10/p{i}", s, std=0.5))) for i, s in enumerate(((7, 5), (33,), (4, 3, 2)))]
        opt = torch.optim.SGD(ps, lr=1e-2, weight_decay=wd, momentum=0.9, nesterov=True)
        for step in range(5):
            opt.zero_grad()
            for i, p in enumerate(ps):
                if i == 1 and step in (0, 1):
                    p.grad = None                       # its momentum buffer is created at the first step that has a gradient
                else:
                    p.grad = torch.from_numpy(fill.fill(f"g10/g{i}/t{step}", tuple(p.shape), std=0.1))
            opt.step()
        for i, p in enumerate(ps):
            out[f"sgd/wd{wd}/p{i}"] = p.detach().numpy().copy()
    save("g10_clip_objective", **out)


# ----------------------------------------------------------------------------------------------- G11 CNN28 (N4)
def g11():
    m = RefCNN28(bias=True, clf=False)
    omodels.deterministic_init(m, tag="cnn28")
    batches = []
    for i in range(4):
        imgs, lbls = otrainer.synthetic_batch(f"g11/b{i}", 8, 8, 28)
        batches.append((imgs[:, :1].contiguous(), lbls))               # 1-channel 28x28
    losses, scores, first = run_trajectory(m, batches, "hsc", lr=1e-3, wd=0.0)
    save("g11_cnn28_hsc", losses=losses, scores=scores, **first)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g7", "g8", "g9", "g10", "g11"]
    for w in which:
        globals()[w]()
