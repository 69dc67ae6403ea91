"""GPU tier 3b: the trainer counterpart (train_cls / eval_cls / run) against the golden trajectories and the
oracle's restatement of the same loop; size-independent properties at the full benchmark batch."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from gpu_util import rel_rms   # noqa: E402
from oracle import models as omodels, trainer as otrainer   # noqa: E402


@pytest.fixture(autouse=True)
def _restore_dtype():
    import eoe_amd
    old = eoe_amd.compute_dtype()
    yield
    eoe_amd.set_compute_dtype(old)


def test_train_cls_matches_golden_trajectory(golden):
    import eoe_amd
    from eoe_amd.data import ListSource
    from eoe_amd.models import ClipViTB32Custom
    from eoe_amd.training import TRAINER
    eoe_amd.set_compute_dtype("fp16")
    g = golden("g3_vit_l2_hsc")
    m = omodels.deterministic_init(ClipViTB32Custom(layers=2), tag="vit", layers=2)
    batches = [otrainer.synthetic_batch(f"g3/b{i}", 2, 2, 224) for i in range(3)]
    tr = TRAINER["hsc"](m, dataset=ListSource(batches), epochs=1, lr=1e-4, wdk=1e-3, milestones=[], batch_size=2)
    model, roc = tr.train_cls(m, tr.ds, 0, "0", 0)
    dev = np.abs(np.array(tr.last_losses) - g["losses"]) / np.maximum(1, np.abs(g["losses"]))
    assert dev.max() < 1e-3, (tr.last_losses, g["losses"])
    # epoch AUC over the 3 step batches equals the oracle's on the golden scores
    from oracle import metrics as ometrics
    la = np.concatenate([b[1].numpy() for b in batches])
    want = ometrics.roc_auc(la, g["scores"].reshape(-1))
    assert abs(roc.auc - want) < 1e-3
    assert not next(model.parameters()).is_cuda and not model.training      # returned on the CPU in eval mode (:471)


@pytest.mark.parametrize("obj,fixture,clf", [("hsc", "g3_vit_l12_hsc_big", False), ("bce", "g3_vit_l12_bce_big", True)])
def test_train_cls_at_the_benchmark_batch_matches_the_reference_trajectory(golden, obj, fixture, clf):
    """the TRAINER (not a loop written in the test) on the benchmark's model at the benchmark's batch: `ADTrainer.train_cls` with its
    defaults -- fp16 gradient-scale policy, backward hand-over buffers, asynchronous weight gradients, the last block on its class-token
    rows, eager launches -- over the K = 10 reference-made step batches of the 12-layer ViT-B/32 (128 + 128 images, a new batch every
    step; ad_trainer.py:406-455).  Loss, scores and the per-batch AUC within the stated 1e-3 of the reference's trajectory on every
    step; next to each AUC deviation the share of pairs the step's own score deviation leaves undecided (parity_util.auc_flip_share)"""
    import eoe_amd
    import parity_util
    from eoe_amd import ops
    from eoe_amd.data import ListSource
    from eoe_amd.models import ClipViTB32Custom
    from eoe_amd.training import TRAINER
    eoe_amd.set_compute_dtype("fp16")
    g = golden(fixture)
    K = len(g["losses"])
    m = omodels.deterministic_init(ClipViTB32Custom(layers=12, clf=clf), tag="vit", layers=12)
    batches = [otrainer.synthetic_batch(f"g3big/b{i}", 128, 128, 224) for i in range(K)]
    tr = TRAINER[obj](m, dataset=ListSource(batches), epochs=1, lr=1e-4, wdk=1e-3, milestones=[], batch_size=128)
    assert ops.VIT_ASYNC_WGRAD and ops.VIT_HANDOVER and ops.VIT_CLS_ONLY_LAST and not tr.graph_steps     # the defaults are what is pinned
    model, roc = tr.train_cls(m, tr.ds, 0, "0", 0)
    labels, scores = tr.last_scores[0]
    labels, scores = labels.cpu().numpy().reshape(K, 256), scores.float().cpu().numpy().reshape(K, 256)
    assert (labels == batches[0][1].numpy()[None, :]).all()
    dl, ds = parity_util.trajectory_deviation(tr.last_losses, list(scores), g)
    aucs = np.array([abs(parity_util.auc_of(labels[k], scores[k]) - parity_util.auc_of(labels[k], g["scores"][k])) for k in range(K)])
    flips = np.array([parity_util.auc_flip_share(labels[k], g["scores"][k], max(float(ds[k]), 1e-12)) for k in range(K)])
    fmt = lambda a: "[" + " ".join(f"{v:.1e}" for v in a) + "]"          # noqa: E731
    print(f"\n[train_cls {obj}, 12 layers, 128 + 128] loss dev {fmt(dl)}\n   score dev {fmt(ds)}\n   AUC dev   {fmt(aucs)}\n"
          f"   pairs within 2 x the step's score deviation (the AUC they could move) {fmt(flips)}; gradient-scale events {tr.scale_events}")
    # (BCE, config 5: the loss spike 0.73 -> 9.7 -> 6.0 at the start carries the forward's logit error; test_gpu_parity_big.py holds the
    #  same run to 2e-3 on the loss there and to 1e-3 on the first three steps: VIT_BCE_BARS)
    loss_bar = 1e-3 if obj == "hsc" else 2e-3
    assert dl.max() <= loss_bar and dl[:3].max() <= 1e-3 and ds.max() <= 1e-3, (fmt(dl), fmt(ds))
    assert aucs.max() <= 1e-3, (fmt(aucs), fmt(flips))
    # the epoch AUC the trainer returns = the oracle's metric on the reference's scores of all K batches
    from oracle import metrics as ometrics
    want = ometrics.roc_auc(labels.reshape(-1), g["scores"].reshape(-1))
    assert abs(roc.auc - want) < 1e-3, (roc.auc, want)


def test_run_loop_and_eval_determinism(tmp_path):
    import eoe_amd
    from eoe_amd.data import SyntheticAD
    from eoe_amd.models import ClipViTB32Custom
    from eoe_amd.training import TRAINER, ADTrainer
    from eoe_amd.training.ad_trainer import JsonLogger
    torch.manual_seed(0)
    ds = SyntheticAD(n_train_normal=48, n_oe=16, n_test=32, res=224, shift=1.0, seed=1, normalize=([0.1, 0.0, -0.1], [1.0, 2.0, 0.5]))
    m = ClipViTB32Custom(layers=1)
    tr = TRAINER["hsc"](m, dataset=ds, epochs=2, lr=1e-4, wdk=1e-3, milestones=[1], batch_size=16,
                        logger=JsonLogger(str(tmp_path)), classes=["only"])
    ADTrainer.KEEP_SNAPSHOT_IN_RAM = True
    try:
        models, res = tr.run(run_seeds=2)
    finally:
        ADTrainer.KEEP_SNAPSHOT_IN_RAM = False
    assert set(res) == {"mean_auc", "mean_avg_prec", "std_auc", "cls_aucs"}
    assert len(res["cls_aucs"]) == 1 and len(res["cls_aucs"][0]) == 2
    assert all(0.0 <= a <= 1.0 for a in res["cls_aucs"][0])
    assert len(tr.last_losses) == 2 * 3 and all(np.isfinite(tr.last_losses))
    # "re-evaluating completed class-seed pairs should yield the same metrics again" (main/__init__.py:120-130)
    roc1, prc1 = tr.eval_cls(models[0][1], ds, 0, "only", 1)
    roc2, prc2 = tr.eval_cls(models[0][1], ds, 0, "only", 1)
    assert roc1.auc == roc2.auc == res["cls_aucs"][0][1] and prc1.avg_prec == prc2.avg_prec
    # the snapshot has the reference's layout and reloads into the oracle model (same names)
    snap = torch.load(str(tmp_path / "snapshots" / "snapshot_cls0_it1.pt"))
    assert set(snap) >= {"net", "opt", "sched", "epoch"} and snap["epoch"] == 2
    ref = omodels.ClipViTNet(layers=1)
    ref.load_state_dict(snap["net"], strict=True)
    x = ds.test_x[:4]
    xn = (x - torch.tensor([0.1, 0.0, -0.1]).view(1, 3, 1, 1)) / torch.tensor([1.0, 2.0, 0.5]).view(1, 3, 1, 1)
    with torch.no_grad():
        want = ref(xn)
        mm = models[0][1].cuda()
        got = mm(x.cuda())          # the fused normalise installed by the trainer is still set on the encoder
    assert rel_rms(got, want) < 3e-3


def test_full_batch_properties():
    """at the benchmark's full step batch (128 + 128): fp16 and bf16 paths agree (no fp16 gradient underflow),
    the loss equals the mean of the per-sample losses recomputed from the returned features, per-sample
    independence (scores of a sample do not depend on its batch mates), and a step changes every parameter."""
    import eoe_amd
    from eoe_amd.models import ClipViTB32Custom
    torch.manual_seed(0)
    n = 256
    x = torch.randn(n, 3, 224, 224, device="cuda")
    y = torch.cat([torch.zeros(n // 2, dtype=torch.long), torch.ones(n // 2, dtype=torch.long)]).cuda()
    m = ClipViTB32Custom(layers=12).cuda().train()
    res = {}
    for dt in ("fp16", "bf16"):
        eoe_amd.set_compute_dtype(dt)
        for p in m.parameters():
            p.grad = None
        f = m(x)
        loss = eoe_amd.hsc_loss(f, y, 0)
        loss.backward()
        res[dt] = (loss.item(), f.detach().clone(), {k: p.grad.double().norm().item() for k, p in m.named_parameters()})
    l16, f16, g16 = res["fp16"]
    lb, fb, gb = res["bf16"]
    assert abs(l16 - lb) < 5e-3 * max(1, abs(l16)), (l16, lb)
    worst = max(abs(g16[k] - gb[k]) / max(gb[k], 1e-30) for k in g16)
    print(f"[full batch] loss fp16 {l16:.6f} bf16 {lb:.6f}; worst grad-norm rel diff fp16 vs bf16 {worst:.2e}")
    assert worst < 3e-2, worst
    # loss = mean of per-sample losses from the same features (fp64 on the host)
    from oracle import objectives
    want = objectives.hsc_loss(f16.double().cpu(), y.cpu()).item()
    assert abs(l16 - want) < 1e-5 * max(1, abs(want))
    # per-sample independence: the first 8 samples alone give the same features
    eoe_amd.set_compute_dtype("fp16")
    with torch.no_grad():
        f8 = m(x[:8])
    assert torch.equal(f8, f16[:8]) or rel_rms(f8, f16[:8].cpu()) < 1e-6
    # one optimiser step moves every trainable tensor
    before = {k: p.detach().clone() for k, p in m.named_parameters()}
    opt = eoe_amd.FusedAdam(m.parameters(), lr=1e-4, weight_decay=1e-3)
    opt.step()
    assert all((before[k] != p.detach()).any().item() for k, p in m.named_parameters())


@pytest.mark.parametrize("objective", ["hsc", "bce"])
def test_trainer_graph_steps_equals_eager(objective):
    """ADTrainer(graph_steps=True): the full-size step batches are replayed from a HIP graph, the ragged last batch runs
    eagerly -- same losses and AUC as the eager trainer (CNN32 at 32x32, the launch-bound configuration)"""
    import copy
    from eoe_amd.data import SyntheticAD
    from eoe_amd.models import CNN32
    from eoe_amd.training import TRAINER
    torch.manual_seed(1)
    m0 = CNN32(bias=True, clf=(objective == "bce"))
    out = {}
    for graph in (False, True):
        torch.manual_seed(5)                                 # same data and loader shuffles in both runs
        ds = SyntheticAD(n_train_normal=40, n_oe=16, n_test=32, res=32, shift=1.0, seed=2, normalize=([0.1, 0.0, -0.1], [1.0, 2.0, 0.5]))
        tr = TRAINER[objective](copy.deepcopy(m0), dataset=ds, epochs=2, lr=1e-3, wdk=0.0, milestones=[], batch_size=16,
                                classes=["only"], graph_steps=graph, exact_bn=False)        # (the 16-bit path: bitwise replay is its property)
        model, roc = tr.train_cls(copy.deepcopy(m0), ds, 0, "only", 0)
        out[graph] = (list(tr.last_losses), roc.auc)
    (le, ae), (lg, ag) = out[False], out[True]
    assert len(le) == len(lg) == 2 * 3                      # 2 full batches + 1 ragged batch per epoch
    # the CNN32 step is free of atomics (bitwise reproducible): replayed and eager steps agree exactly for HSC; the BCE head's
    # 1-wide linear uses fp32 atomics in its weight gradient, so that trajectory is only close
    if objective == "hsc":
        assert lg == le, (lg, le)
    np.testing.assert_allclose(lg, le, rtol=1e-2)
    assert abs(ae - ag) < 2e-2
