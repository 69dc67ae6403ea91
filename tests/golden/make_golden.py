#!/usr/bin/env python
"""Generates the golden fixtures under tests/golden/ by running the REFERENCE's own modules.

Run in the build container only (needs /root/reference; never on the GPU box):
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What is reference code here (all executed from /root/reference/src, nothing re-typed):
  * models: `eoe.models.cnn.CNN32/CNN28`, `eoe.models.custom_base.CustomNet`, `eoe.models.resnet.WideResNet`,
    `eoe.models.cbam.CBAM`, and `clip/model.py`'s `VisualTransformer` / `ResidualAttentionBlock`;
  * objectives: the `loss` / `compute_anomaly_score` / `prepare_metric` methods of `eoe.training.{hsc,bce,dsad,
    dsvdd,focal,clip}` -- the trainer classes are loaded by file path with `eoe.training.ad_trainer.ADTrainer`
    replaced by an empty base class (the real one drags in torchvision / kornia / cv2 / tensorboard, which are
    absent here: ordinary ModuleNotFoundError) and instantiated without `__init__`;
  * the step-batch layout: `eoe.datasets.bases.BalancedConcatLoader`, loaded the same way, over stock
    `torch.utils.data.DataLoader`s;
  * the optimiser / scheduler / metrics are the stock third-party calls the reference's trainer makes
    (`torch.optim.Adam`, `torch.optim.SGD`, `MultiStepLR`, `sklearn.metrics.roc_curve/auc/average_precision_score`;
    `src/eoe/training/ad_trainer.py:8,380-384,453-454,517-521`).
Weights and inputs come from oracle.fill (a pure function of name/shape), so the fixtures hold only small
outputs.  Only data is written; no reference source is copied.

The `*_big` fixtures are the well-conditioned parity cases (SURVEY.md section 8d "Parity run": K = 10 steps at the
benchmark batch of 128 + 128 images for the BatchNorm nets, 16 + 16 for WideResNet at 224 x 224, and one full
step of the 12-layer ViT-B/32 at 128 + 128 = the benchmark's M = 12 800 token rows).
"""
import importlib.util
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.environ.get("EOE_GOLDEN_OUT", HERE)     # write somewhere else to compare against the committed files
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/src")
sys.dont_write_bytecode = True

from oracle import fill, models as omodels, trainer as otrainer  # noqa: E402  (only for the fill rule / batches)

import types                                                       # noqa: E402

REF = "/root/reference/src/eoe"


def _stub(name, **attrs):
    """an empty stand-in module for an import the loaded file makes but the code under test never touches"""
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    if "." in name:
        parent, leaf = name.rsplit(".", 1)
        if parent in sys.modules:
            setattr(sys.modules[parent], leaf, m)
    return m


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


from eoe.models.cnn import CNN32 as RefCNN32, CNN28 as RefCNN28    # noqa: E402
from eoe.models.custom_base import CustomNet as RefCustomNet       # noqa: E402

ref_clip = _load("ref_clip_model", f"{REF}/models/clip_official/clip/model.py")

# absent third-party packages and the reference modules that need them: empty stand-ins (never called)
_stub("torchvision")
_stub("torchvision.transforms", Compose=object)
_stub("torchvision.datasets", VisionDataset=object)
_stub("torchvision.models", wide_resnet50_2=None)      # resnet.py:3, used only by the out-of-scope WideResNet50Pretrained
_stub("eoe.models.clip_official").__path__ = []
_stub("eoe.models.clip_official.clip")                 # clip.py:2 (checkpoint loader; not used by loss / score)
_stub("eoe.datasets", str_labels=None).__path__ = []
_stub("eoe.utils").__path__ = []
_stub("eoe.utils.logger", Logger=object)
_stub("eoe.utils.transformations", ConditionalCompose=object, GPU_TRANSFORMS={}, Normalize=object,
      GlobalContrastNormalization=object)
_load("eoe.utils.stats", f"{REF}/utils/stats.py")
_stub("eoe.training").__path__ = []
_stub("eoe.training.ad_trainer", ADTrainer=type("ADTrainer", (), {}))     # ad_trainer.py:93 (base class only)
np.infty = np.inf                                       # bases.py:82 uses the numpy < 2 alias


def ref_trainer(name, cls, **attrs):
    """an instance of the reference's trainer class `cls` from training/<name>.py WITHOUT running ADTrainer.__init__
    (which builds loggers and datasets): only the objective methods are used"""
    mod = sys.modules.get(f"eoe.training.{name}") or _load(f"eoe.training.{name}", f"{REF}/training/{name}.py")
    obj = object.__new__(getattr(mod, cls))
    obj.__dict__.update(device=torch.device("cpu"), **attrs)
    return obj


HSC = ref_trainer("hsc", "HSCTrainer")
BCE = ref_trainer("bce", "BCETrainer")
DSAD = ref_trainer("dsad", "DSADTrainer")
DSVDD = ref_trainer("dsvdd", "DSVDDTrainer")
FOCAL = ref_trainer("focal", "FocalTrainer")
ref_bases = _load("eoe.datasets.bases", f"{REF}/datasets/bases.py")

torch.set_num_threads(8)
torch.manual_seed(0)


def hsc_loss_ref(f, y):
    return HSC.loss(f, y, None, nominal_label=0)                    # training/hsc.py:17-21


def hsc_score_ref(f):
    return HSC.compute_anomaly_score(f, None, nominal_label=0)     # training/hsc.py:12-15


def bce_loss_ref(f, y):
    return BCE.loss(f, y, None, nominal_label=0)                    # training/bce.py:19-20


def bce_score_ref(f):
    return BCE.compute_anomaly_score(f, None, nominal_label=0)     # training/bce.py:15-17


LOSS = {"hsc": (hsc_loss_ref, hsc_score_ref), "bce": (bce_loss_ref, bce_score_ref)}


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
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
def dsad_loss_ref(f, y, nominal=0):
    return DSAD.loss(f, y, None, nominal_label=nominal)             # training/dsad.py:17-21


def dsvdd_loss_ref(f, c):
    return DSVDD.loss(f, None, c)                                   # training/dsvdd.py:26-27


def dsvdd_center_ref(batch_feats, eps=1e-1):
    """training/dsvdd.py:10-22 (prepare_metric) over a loader whose "images" are already the nominal features and
    an identity model"""
    loader = [(bf, torch.zeros(bf.shape[0], dtype=torch.long), None) for bf in batch_feats]
    return DSVDD.prepare_metric("0", loader, torch.nn.Identity(), 0, eps=eps)


def focal_loss_ref(x, y):
    return FOCAL.loss(x, y, None)                                   # training/focal.py:11-24,34-36


def g9():
    f = torch.from_numpy(fill.fill("g9/features", (16, 256), std=0.08))
    y = torch.from_numpy(fill.fill_int("g9/labels", (16,), 0, 2))
    out = {}
    ff = f.clone().requires_grad_(True)
    loss = dsad_loss_ref(ff, y)
    loss.backward()
    out["dsad_loss"], out["dsad_grad"] = loss.item(), ff.grad.numpy()
    out["dsad_scores"] = DSAD.compute_anomaly_score(f, None).numpy()                # dsad.py:12-15
    feats = [torch.from_numpy(fill.fill(f"g9/cb{i}", (5 + i, 256), std=0.3, mean=0.02)) for i in range(3)]
    c = dsvdd_center_ref(feats)
    out["dsvdd_center"] = c.numpy()
    ff = f.clone().requires_grad_(True)
    loss = dsvdd_loss_ref(ff, c)
    loss.backward()
    out["dsvdd_loss"], out["dsvdd_grad"] = loss.item(), ff.grad.numpy()
    out["dsvdd_scores"] = DSVDD.compute_anomaly_score(f, c).numpy()                 # dsvdd.py:23-24
    x = torch.cat([f[:, :1] * 20, torch.tensor([[40.0], [-40.0], [0.0], [18.0]])])     # incl. saturated logits (pt clamp)
    yy = torch.cat([y, torch.tensor([1, 0, 1, 0])])
    xx = x.clone().requires_grad_(True)
    loss = focal_loss_ref(xx, yy)
    loss.backward()
    out["focal_x"], out["focal_y"] = x.numpy(), yy.numpy()
    out["focal_loss"], out["focal_grad"] = loss.item(), xx.grad.numpy()
    out["focal_scores"] = FOCAL.compute_anomaly_score(x, None, nominal_label=0).numpy()   # focal.py:30-32
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
def clip_loss_ref(image_features, labels, text_features, nominal_label=0, ad_mode="one_vs_rest"):
    t = ref_trainer("clip", "ADClipTrainer", ad_mode=ad_mode)
    return t.loss(image_features, labels, text_features, nominal_label=nominal_label)              # training/clip.py:81-103


def clip_score_ref(image_features, center, ad_mode="one_vs_rest"):
    t = ref_trainer("clip", "ADClipTrainer", ad_mode=ad_mode)
    return t.compute_anomaly_score(image_features, center)                                          # training/clip.py:66-79


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


# ----------------------------------------------------------------------------------------------- G12 step-batch layout (A0)
class _IndexedSet(torch.utils.data.Dataset):
    """(image, label, index) triples like the reference's datasets return (bases.py:543-564); the 'image' encodes
    (set id, index) so that the fixture shows which sample landed where"""

    def __init__(self, n, label, set_id):
        self.n, self.label, self.set_id = n, label, set_id

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return torch.tensor([float(self.set_id), float(i)]), self.label, i


def g12():
    """`BalancedConcatLoader` of the reference (bases.py:570-600) over stock DataLoaders: ragged last batch, OE set smaller
    than the normal set (index list tiled, :580-584), OE batches that must be concatenated to reach the normal batch's
    size (:593-594), OE index offset by the length of the normal *dataset* (not subset; :596)"""
    from torch.utils.data import DataLoader, Subset
    out = {}
    cases = {  # name: (normal dataset size, normal subset, oe dataset size, oe subset, normal batch, oe batch)
        "ragged": (20, list(range(3, 16)), 9, [1, 4, 5, 7], 5, 5),
        "oe_larger": (12, list(range(12)), 40, list(range(5, 37)), 4, 4),
        "oe_small_batches": (16, [0, 2, 4, 6, 8, 10, 12, 14, 15], 30, list(range(30)), 4, 3),
        "single_oe": (10, list(range(10)), 5, [3], 4, 4),
    }
    for name, (nn_, nsub, no, osub, nb, ob) in cases.items():
        nd, od = Subset(_IndexedSet(nn_, 0, 0), list(nsub)), Subset(_IndexedSet(no, 1, 1), list(osub))
        loader = ref_bases.BalancedConcatLoader(DataLoader(nd, batch_size=nb, shuffle=False),
                                                DataLoader(od, batch_size=ob, shuffle=False))
        out[f"{name}/cfg"] = np.array([nn_, no, nb, ob], np.int64)
        out[f"{name}/normal_subset"], out[f"{name}/oe_subset"] = np.array(nsub, np.int64), np.array(osub, np.int64)
        out[f"{name}/oe_indices_tiled"] = np.array(od.indices, np.int64)
        out[f"{name}/len"] = len(loader)
        for b, (imgs, lbls, idcs) in enumerate(loader):
            out[f"{name}/b{b}/imgs"], out[f"{name}/b{b}/lbls"], out[f"{name}/b{b}/idcs"] = imgs.numpy(), lbls.numpy(), idcs.numpy()
        out[f"{name}/n_batches"] = b + 1
    save("g12_batching", **out)


def g15():
    """the class x seed loop's task definition: `ADTrainer.get_nominal_classes` (training/ad_trainer.py:166-175; the module cannot
    be imported -- torchvision / kornia / cv2 --, so the method's own definition is taken out of the file with `ast` and executed)
    for every class of 3-, 10- and 30-class sets under the three AD modes, and `TorchvisionDataset.create_subset`
    (datasets/bases.py:169-203) = the rows of a labelled split that form the normal training set"""
    import ast
    import types as _types
    src = open(f"{REF}/training/ad_trainer.py").read()
    cls = next(n for n in ast.parse(src).body if isinstance(n, ast.ClassDef) and n.name == "ADTrainer")
    fn = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "get_nominal_classes")
    ns = {"no_classes": lambda ds: ds, "ADTrainer": _types.SimpleNamespace(AD_MODES=("one_vs_rest", "leave_one_out", "fifty_fifty"))}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), f"{REF}/training/ad_trainer.py", "exec"), ns)
    out = {}
    for n in (3, 10, 30):
        for mode in ("one_vs_rest", "leave_one_out", "fifty_fifty"):
            for c in range(n):
                me = _types.SimpleNamespace(ad_mode=mode, dsstr=n)
                out[f"nominal/{n}/{mode}/{c}"] = np.array(ns["get_nominal_classes"](me, c), np.int64)
    from torch.utils.data import Dataset

    class _Split(Dataset):
        def __len__(self):
            return 0
    rng = np.random.RandomState(3)
    for name, (n_cls, n_rows) in {"small": (3, 17), "cifar_like": (10, 200), "in30_like": (30, 390)}.items():
        labels = rng.randint(0, n_cls, n_rows).astype(np.int64)
        out[f"subset/{name}/labels"] = labels
        for mode, normal in (("ovr", [1]), ("loo", [c for c in range(n_cls) if c != 1]), ("ff", list(range(1, n_cls // 2 + 1)))):
            me = _types.SimpleNamespace(normal_classes=tuple(normal), limit_samples=np.inf)
            sub = ref_bases.TorchvisionDataset.create_subset(me, _Split(), labels.tolist())
            out[f"subset/{name}/{mode}/normal_classes"] = np.array(normal, np.int64)
            out[f"subset/{name}/{mode}/indices"] = np.array(sub.indices, np.int64)
    save("g15_tasks", **out)


def g16():
    """fp16-weights mode: the reference's own `convert_weights` (clip/model.py:371-392) applied to its own
    VisualTransformer -- which parameters become fp16 tensors (by state_dict name) -- and, on a few of those tensors, the values
    `.half()` leaves; plus five steps of torch.optim.SGD(momentum 0.9, nesterov, weight decay) as the reference constructs it for CLIP
    models (ad_trainer.py:380-381) on fp16 CPU tensors with fp16 gradients (the CPU kernels round the `alpha` of add(x, alpha=...) to fp16
    first, the GPU kernels keep it in fp32: tests/test_gpu_round3.py pins the GPU form against stock torch on the device, this fixture
    pins the op order and the rounding points)"""
    vt = ref_clip.VisualTransformer(input_resolution=64, patch_size=32, width=128, layers=2, heads=2, output_dim=32)
    ref_clip.convert_weights(vt)
    names = sorted(vt.state_dict().keys())
    out = {"names": np.array(names), "is_fp16": np.array([vt.state_dict()[n].dtype == torch.float16 for n in names])}
    torch.manual_seed(16)
    p0 = torch.randn(4096).half()
    grads = [(torch.randn(4096) * (0.5 + i)).half() for i in range(5)]
    q = torch.nn.Parameter(p0.clone())
    opt = torch.optim.SGD([q], lr=1e-2, momentum=0.9, nesterov=True, weight_decay=1e-3)
    for g in grads:
        q.grad = g.clone()
        opt.step()
    out["sgd/p0"], out["sgd/grads"] = p0.float().numpy(), torch.stack(grads).float().numpy()
    out["sgd/p5"], out["sgd/buf5"] = q.detach().float().numpy(), opt.state[q]["momentum_buffer"].float().numpy()
    save("g16_fp16_weights", **out)


# ----------------------------------------------------------------------------------------------- big, well-conditioned parity cases
def run_trajectory_big(model, batches_fn, n_steps, objective, lr, wd, twin64=True):
    """as run_trajectory with batches produced one at a time (memory), run twice: the reference modules in fp32 (the
    fixture proper) and the SAME modules cast to fp64 on the same inputs.  |fp32 - fp64| is the reference's own rounding
    noise on this trajectory: where it exceeds the 1e-3 parity bar, two correct fp32 implementations (or the reference
    on another BLAS) already disagree by more than the bar, so the tests scale their tolerance by it"""
    import copy

    def gen(double):
        for i in range(n_steps):
            imgs, lbls = batches_fn(i)
            yield (imgs.double() if double else imgs), lbls
    m64 = copy.deepcopy(model).double() if twin64 else None
    losses, scores, first = run_trajectory(model, gen(False), objective, lr, wd)
    if not twin64:
        return losses, scores, first
    l64, s64, f64 = run_trajectory(m64, gen(True), objective, lr, wd)
    first["losses64"], first["scores64"] = l64, s64
    for k in list(f64):
        if k.startswith("gnorm/"):
            first["gnorm64/" + k[6:]] = f64[k]
    return losses, scores, first


def g2big():
    """CNN32 at the benchmark batch (128 + 128, train_cifar.py:20), K = 10 Adam steps"""
    for clf, obj in ((False, "hsc"), (True, "bce")):
        m = RefCNN32(bias=True, clf=clf)
        omodels.deterministic_init(m, tag="cnn32")
        losses, scores, first = run_trajectory_big(m, lambda i: otrainer.synthetic_batch(f"g2big/b{i}", 128, 128, 32), 10,
                                                   obj, lr=1e-3, wd=0.0)
        save(f"g2_cnn32_{obj}_big", losses=losses, scores=scores, **first)


def g11big():
    m = RefCNN28(bias=True, clf=False)
    omodels.deterministic_init(m, tag="cnn28")

    def batch(i):
        imgs, lbls = otrainer.synthetic_batch(f"g11big/b{i}", 128, 128, 28)
        return imgs[:, :1].contiguous(), lbls
    losses, scores, first = run_trajectory_big(m, batch, 10, "hsc", lr=1e-3, wd=0.0)
    save("g11_cnn28_hsc_big", losses=losses, scores=scores, **first)


def g5big():
    """WideResNet + CBAM at 16 + 16 images of 224 x 224, K = 10 Adam steps (train_imagenet.py:16-17 lr / wd)"""
    from eoe.models.resnet import WideResNet as RefWRN
    m = RefWRN()
    omodels.deterministic_init(m, tag="wrn")
    losses, scores, first = run_trajectory_big(m, lambda i: otrainer.synthetic_batch(f"g5big/b{i}", 16, 16, 224), 10,
                                               "hsc", lr=1e-3, wd=0.0)
    save("g5_wideresnet_hsc_big", losses=losses, scores=scores, **first)


def g5full():
    """WideResNet + CBAM at the FULL benchmark batch (128 + 128 images of 224 x 224: every convolution at its benchmark geometry,
    M = 802 816 rows in layer1), K = 10 Adam steps, with the fp64 twin"""
    from eoe.models.resnet import WideResNet as RefWRN
    m = RefWRN()
    omodels.deterministic_init(m, tag="wrn")
    losses, scores, first = run_trajectory_big(m, lambda i: otrainer.synthetic_batch(f"g5full/b{i}", 128, 128, 224), 10,
                                               "hsc", lr=1e-3, wd=0.0)
    save("g5_wideresnet_hsc_full", losses=losses, scores=scores, **first)


def g13():
    """BASELINE.json config 2, "WideResNet backbone, 32 x 32": the reference's WideResNet accepts 224 x 224 only (resnet.py:86,38),
    so the fixture drives the reference's OWN layers (conv1 / bn1 / maxpool / layer1-4 with their BasicBlock + CBAM / fc,
    resnet.py:33-48,112-149) in the reference's order (resnet.py:87-107) on 32 x 32 inputs; the only departure is the final
    pooling, which averages the 1 x 1 map that is left instead of AvgPool2d(7).  128 + 128 images, K = 10 Adam steps."""
    from eoe.models.resnet import WideResNet as RefWRN

    class RefWRN32(RefWRN):
        def forward(self, x):
            x = self.maxpool(self.relu(self.bn1(self.conv1(x.view(-1, 3, 32, 32)))))
            x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
            x = x.mean((2, 3))
            x = self.fc(x)
            return self.linear(x) if self.clf else x
    m = RefWRN32()
    omodels.deterministic_init(m, tag="wrn")
    losses, scores, first = run_trajectory_big(m, lambda i: otrainer.synthetic_batch(f"g13/b{i}", 128, 128, 32), 10,
                                               "hsc", lr=1e-3, wd=0.0)
    save("g13_wideresnet32_hsc", losses=losses, scores=scores, **first)


def g14():
    """Resize / ColorJitter / CLIP preprocessing as the reference's runners apply them to PIL images (train_imagenet.py:31,
    train_clip_imagenet.py:28-29, train_cifar.py:32, clip_official/clip/clip.py:58-65).  torchvision is absent here; its PIL code
    path is a thin wrapper over Pillow (functional_pil: Image.resize, ImageEnhance.Brightness / Contrast / Color, HSV round
    trip with `np_h += np.uint8(hue_factor * 255)`), so the fixture calls Pillow -- the third-party arithmetic -- directly."""
    from PIL import Image, ImageEnhance
    import PIL
    out = {"pillow_version": np.array(PIL.__version__)}
    rng_img = (fill.fill("g14/img", (3, 75, 100, 3), std=0.6, mean=0.0) * 128 + 128).clip(0, 255).astype(np.uint8)
    rng_img[0, :20, :30] = 0
    rng_img[0, 20:30, :10] = 255
    rng_img[1, 40:50, 50:70] = rng_img[1, 40:50, 50:70, :1]                   # a gray patch (s = 0 in HSV)
    out["images"] = rng_img
    for name, size, filt in (("bilinear_64x48", (64, 48), Image.BILINEAR), ("bilinear_32x32", (32, 32), Image.BILINEAR),
                             ("bilinear_150x200", (150, 200), Image.BILINEAR), ("bicubic_64x48", (64, 48), Image.BICUBIC),
                             ("bicubic_150x200", (150, 200), Image.BICUBIC)):
        out[f"resize/{name}"] = np.stack([np.asarray(Image.fromarray(im).resize((size[1], size[0]), filt)) for im in rng_img])
    # torchvision Resize(int): shorter side -> size, longer side int(size * long / short)
    out["resize/bicubic_short56"] = np.stack([np.asarray(Image.fromarray(im).resize((int(56 * 100 / 75), 56), Image.BICUBIC)) for im in rng_img])

    def tv_hue(img, hue_factor):                          # torchvision.transforms.functional_pil.adjust_hue
        h, s, v = img.convert("HSV").split()
        np_h = np.array(h, dtype=np.uint8)
        with np.errstate(over="ignore"):
            np_h += np.uint8(int(hue_factor * 255) & 0xFF)
        return Image.merge("HSV", (Image.fromarray(np_h, "L"), s, v)).convert("RGB")

    ops = [lambda im, f: ImageEnhance.Brightness(im).enhance(f), lambda im, f: ImageEnhance.Contrast(im).enhance(f),
           lambda im, f: ImageEnhance.Color(im).enhance(f), tv_hue]
    factors = np.array([[0.993, 1.008, 0.991, 0.0071], [1.0095, 0.9902, 1.0049, -0.0093], [0.7, 1.6, 0.4, 0.31]], np.float32)
    orders = np.array([[0, 1, 2, 3], [3, 1, 0, 2], [2, 3, 1, 0]], np.int32)
    jit = []
    for im, f, o in zip(rng_img, factors, orders):
        pim = Image.fromarray(im)
        for op in o:
            pim = ops[int(op)](pim, float(f[int(op)]))
        jit.append(np.asarray(pim))
    out["jitter/factors"], out["jitter/orders"], out["jitter/out"] = factors, orders, np.stack(jit)
    # CLIP _transform(32) on the 75 x 100 images: Resize(32, bicubic) -> CenterCrop(32) -> ToTensor -> Normalize
    mean = np.array([0.48145466, 0.4578275, 0.40821073], np.float32).reshape(1, 3, 1, 1)
    std = np.array([0.26862954, 0.26130258, 0.27577711], np.float32).reshape(1, 3, 1, 1)
    res = []
    for im in rng_img:
        r = np.asarray(Image.fromarray(im).resize((int(32 * 100 / 75), 32), Image.BICUBIC))
        top, left = int(round((r.shape[0] - 32) / 2.0)), int(round((r.shape[1] - 32) / 2.0))
        res.append(r[top:top + 32, left:left + 32])
    t = torch.from_numpy(np.stack(res)).permute(0, 3, 1, 2).float().div(255)           # ToTensor
    out["clip/out"] = ((t.numpy() - mean) / std).astype(np.float32)
    save("g14_pil_transforms", **out)


def g3big():
    """the 12-layer ViT-B/32 + head, K = 10 full fine-tune steps at the benchmark batch (128 + 128 images = 12 800 token rows, a new
    batch every step; SURVEY.md section 8d's parity run on the headline model): features and per-tensor gradient summaries of the
    first step, loss and scores of every step.  No fp64 twin: the reference's LayerNorm computes in fp32 whatever the input
    (clip/model.py:156-159) and raises on fp64 parameters"""
    m = RefClipNet(12)
    omodels.deterministic_init(m, tag="vit", layers=12)
    losses, scores, first = run_trajectory_big(m, lambda i: otrainer.synthetic_batch(f"g3big/b{i}", 128, 128, 224), 10,
                                               "hsc", lr=1e-4, wd=1e-3, twin64=False)
    save("g3_vit_l12_hsc_big", losses=losses, scores=scores, **first)


def g3long():
    """how far does the 1e-3 bar hold?  the 12-layer ViT at the benchmark batch for K = 40 steps (losses and scores only)"""
    m = RefClipNet(12)
    omodels.deterministic_init(m, tag="vit", layers=12)
    losses, scores, first = run_trajectory_big(m, lambda i: otrainer.synthetic_batch(f"g3big/b{i}", 128, 128, 224), 40,
                                               "hsc", lr=1e-4, wd=1e-3, twin64=False)
    save("g3_vit_l12_hsc_long", losses=losses, scores=scores)


def g3bigfrozen():
    """BASELINE config 4 ("CLIP ViT-B/32 frozen encoder + HSC head"): freeze_parts() on the 12-layer encoder, only the head trains;
    K = 10 steps at the benchmark batch, same batches as g3big, the CLIP runner's lr / wd (train_clip_imagenet.py:13-17)"""
    m = RefClipNet(12, freeze=True)
    omodels.deterministic_init(m, tag="vit", layers=12)

    def gen():
        for i in range(10):
            yield otrainer.synthetic_batch(f"g3big/b{i}", 128, 128, 224)
    losses, scores, first = run_trajectory(m, gen(), "hsc", lr=1e-4, wd=1e-3, freeze=True)
    save("g3_vit_l12_hsc_frozen_big", losses=losses, scores=scores, **first)


def g3bigfrozenlr():
    """config 4 again with a head learning rate at which the ranking means something: with lr 1e-4 the head hardly moves in ten steps,
    every score stays within 1e-5 of 1 and the single-batch AUC ranks float32 ulps.  Adam at lr 1e-2 for K = 80 steps pulls the normal
    half towards the centre (scores spread over (0, 1)); same frozen encoder, the batches b0 .. b79 with the OE half shifted by 3 x the fixed pattern (shift 0.5 leaves the random-weight encoder's
    AUC at 0.5: both halves interleave and any rounding swaps dozens of pairs)"""
    m = RefClipNet(12, freeze=True)
    omodels.deterministic_init(m, tag="vit", layers=12)

    def gen():
        for i in range(80):
            yield otrainer.synthetic_batch(f"g3big/b{i}", 128, 128, 224, shift=3.0)
    losses, scores, first = run_trajectory(m, gen(), "hsc", lr=1e-2, wd=1e-3, freeze=True)
    save("g3_vit_l12_hsc_frozen_lr", losses=losses, scores=scores, **first)


def g3bigbce():
    """BASELINE config 5 ("CLIP ViT-B/32 full fine-tune, BCE"): the 12-layer ViT with the 1-logit head (clf=True) and the BCE objective,
    K = 10 steps at the benchmark batch, same batches as g3big"""
    m = RefClipNet(12, clf=True)
    omodels.deterministic_init(m, tag="vit", layers=12)
    losses, scores, first = run_trajectory_big(m, lambda i: otrainer.synthetic_batch(f"g3big/b{i}", 128, 128, 224), 10,
                                               "bce", lr=1e-4, wd=1e-3, twin64=False)
    save("g3_vit_l12_bce_big", losses=losses, scores=scores, **first)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g7", "g8", "g9", "g10", "g11", "g12",
                             "g2big", "g11big", "g5big", "g5full", "g3big", "g3long", "g3bigfrozen", "g3bigfrozenlr", "g3bigbce", "g13", "g14", "g15", "g16"]
    for w in which:
        globals()[w]()
