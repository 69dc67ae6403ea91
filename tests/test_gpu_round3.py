"""GPU tier (round 3): the non-finite guard of the scaled fp16 step, the dynamic gradient scale of the trainer, and the class x seed
loop over a labelled image set in the reference's AD modes (leave-one-out, one-vs-rest; `training/ad_trainer.py:166-175, 248-253`)."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import batching, models as omodels, trainer as otrainer   # noqa: E402
from gpu_util import rel_rms   # noqa: E402


@pytest.fixture(autouse=True)
def _restore_state():
    import eoe_amd
    old_dt, old_scale = eoe_amd.compute_dtype(), eoe_amd.grad_scale()
    yield
    eoe_amd.set_compute_dtype(old_dt)
    eoe_amd.set_grad_scale(old_scale)


def _params(seed=0):
    g = torch.Generator().manual_seed(seed)
    shapes = [(5,), (2 * 8192 + 3,), (33, 100), (8192,)]
    return [torch.nn.Parameter(torch.randn(s, generator=g).cuda()) for s in shapes]


def _set_grads(ps, seed):
    g = torch.Generator().manual_seed(1000 + seed)
    for p in ps:
        p.grad = torch.randn(p.shape, generator=g).cuda()


@pytest.mark.parametrize("bad", [float("inf"), float("-inf"), float("nan")])
@pytest.mark.parametrize("groups", [1, 2])
def test_adam_drops_a_step_with_a_non_finite_gradient(bad, groups):
    """an injected inf / NaN anywhere (any parameter group) leaves EVERY parameter and moment untouched; the clean steps around
    it are `torch.optim.Adam`'s (the dropped step does not count towards the bias corrections once `skipped_steps()` was read)"""
    import eoe_amd
    ps = _params()
    ref_ps = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    split = [{"params": ps[:2]}, {"params": ps[2:], "lr": 3e-3}] if groups == 2 else ps
    ref_split = [{"params": ref_ps[:2]}, {"params": ref_ps[2:], "lr": 3e-3}] if groups == 2 else ref_ps
    opt = eoe_amd.FusedAdam(split, lr=1e-2, weight_decay=1e-3, guard=True)
    ref = torch.optim.Adam(ref_split, lr=1e-2, weight_decay=1e-3)
    _set_grads(ps, 0)
    for p, q in zip(ps, ref_ps):
        q.grad = p.grad.clone()
    opt.step()
    ref.step()
    before = [(p.detach().clone(), opt.state[p]["exp_avg"].clone(), opt.state[p]["exp_avg_sq"].clone()) for p in ps]
    _set_grads(ps, 1)
    ps[3].grad.view(-1)[4321] = bad                       # last tensor of the LAST group
    opt.step()
    torch.cuda.synchronize()
    for p, (w, m, v) in zip(ps, before):
        assert torch.equal(p.detach(), w) and torch.equal(opt.state[p]["exp_avg"], m) and torch.equal(opt.state[p]["exp_avg_sq"], v)
    assert opt.skipped_steps() == 1 and opt.skipped_steps() == 0
    assert all(int(opt.state[p]["step"].item()) == 1 for p in ps)
    _set_grads(ps, 2)
    for p, q in zip(ps, ref_ps):
        q.grad = p.grad.clone()
    opt.step()
    ref.step()
    for p, q in zip(ps, ref_ps):
        assert torch.isfinite(p).all()
        torch.testing.assert_close(p.detach(), q.detach(), rtol=2e-6, atol=2e-7)
    assert opt.skipped_steps() == 0


def test_sgd_drops_a_step_with_a_non_finite_gradient():
    import eoe_amd
    ps = _params(3)
    opt = eoe_amd.FusedSGD(ps, lr=1e-2, momentum=0.9, nesterov=True, weight_decay=1e-3, guard=True)
    _set_grads(ps, 0)
    opt.step()
    before = [(p.detach().clone(), opt.state[p]["momentum_buffer"].clone()) for p in ps]
    _set_grads(ps, 1)
    ps[0].grad[2] = float("inf")
    opt.step()
    for p, (w, b) in zip(ps, before):
        assert torch.equal(p.detach(), w) and torch.equal(opt.state[p]["momentum_buffer"], b)
    assert opt.skipped_steps() == 1
    _set_grads(ps, 2)
    opt.step()
    assert all(not torch.equal(p.detach(), w) for p, (w, _) in zip(ps, before)) and opt.skipped_steps() == 0


def test_guard_is_off_without_a_scale_and_costs_nothing_then():
    """bf16 / unscaled runs do not pay for the gradient pass: no state is created, inf goes through as in torch.optim.Adam"""
    import eoe_amd
    eoe_amd.set_grad_scale(1.0)
    ps = _params(5)
    opt = eoe_amd.FusedAdam(ps, lr=1e-2)
    _set_grads(ps, 0)
    opt.step()
    assert opt._guard_state is None and opt.skipped_steps() == 0


def test_trainer_halves_an_overflowing_scale(monkeypatch):
    """fp16 with an absurd starting scale: the 16-bit backward chain overflows, the steps are dropped on the device (weights and
    moments stay finite), the trainer halves the scale until steps go through again, and training then proceeds"""
    import eoe_amd
    from eoe_amd import ops
    from eoe_amd.data import SyntheticAD
    from eoe_amd.models import CNN32
    from eoe_amd.training import TRAINER
    eoe_amd.set_compute_dtype("fp16")
    monkeypatch.setattr(ops, "default_grad_scale", lambda dtype=None: 2.0 ** 36)
    torch.manual_seed(3)
    ds = SyntheticAD(n_train_normal=64, n_oe=64, n_test=32, res=32, shift=1.0, seed=2)
    m0 = CNN32(bias=True)
    tr = TRAINER["hsc"](copy.deepcopy(m0), dataset=ds, epochs=8, lr=1e-3, wdk=0.0, milestones=[], batch_size=16, classes=["only"],
                        exact_bn=False)                      # the 16-bit path is the one that can overflow
    tr.SCALE_POLL_EVERY = 1
    model, roc = tr.train_cls(copy.deepcopy(m0), ds, 0, "only", 0)
    assert len(tr.scale_events) >= 4, tr.scale_events                 # several halvings
    assert all(b[1] == a[1] / 2 for a, b in zip(tr.scale_events, tr.scale_events[1:]))
    final = tr.scale_events[-1][1]
    assert 1.0 <= final < 2.0 ** 36
    assert all(torch.isfinite(p).all() for p in model.parameters())
    assert all(np.isfinite(tr.last_losses))
    assert tr.last_losses[-1] < tr.last_losses[0]                     # the steps after the scale settled did train
    assert roc is not None and 0.0 <= roc.auc <= 1.0


# ----------------------------------------------------------------------------------------------- class x seed loop, AD modes
class _Recorder:
    """a step-batch source that remembers what it handed out (for the oracle's run on the SAME splits)"""

    def __init__(self, inner):
        self.inner, self.train_batches, self.test_batches = inner, [], None
        self.nominal_label, self.normalize, self.ds_statistics = inner.nominal_label, inner.normalize, None

    def loaders(self, batch_size, **kw):
        train, test = self.inner.loaders(batch_size, **kw)
        outer = self

        class _T:
            def __iter__(s):
                for b in train:
                    outer.train_batches.append(tuple(t.detach().cpu().clone() for t in b))
                    yield b

            def __len__(s):
                return len(train)
        self.test_batches = [tuple(t.detach().cpu().clone() for t in b) for b in test]
        return _T(), test


def _labelled_set(n_cls=3, per_cls_train=12, per_cls_test=8, n_oe=10, res=224):
    """uint8 images with a class-dependent low-frequency pattern (so that held-out classes ARE anomalous) + noise"""
    g = torch.Generator().manual_seed(11)
    yy, xx = torch.meshgrid(torch.linspace(0, 1, res), torch.linspace(0, 1, res), indexing="ij")
    pats = [torch.stack([torch.sin(6.28 * (k + 1) * xx + k), torch.cos(6.28 * (k + 1) * yy - k), torch.sin(6.28 * (xx + yy) * (k + 1))], -1)
            for k in range(n_cls + 1)]

    def make(k, n):
        x = 128 + 60 * pats[k].unsqueeze(0) + 25 * torch.randn((n, res, res, 3), generator=g)
        return x.clamp(0, 255).to(torch.uint8)
    train = torch.cat([make(k, per_cls_train) for k in range(n_cls)])
    train_y = torch.arange(n_cls).repeat_interleave(per_cls_train)
    perm = torch.randperm(len(train_y), generator=g)                  # classes interleaved, as in a real set
    train, train_y = train[perm], train_y[perm]
    test = torch.cat([make(k, per_cls_test) for k in range(n_cls)])
    test_y = torch.arange(n_cls).repeat_interleave(per_cls_test)
    oe = make(n_cls, n_oe)
    return train, train_y, test, test_y, oe


@pytest.mark.parametrize("ad_mode", ["leave_one_out", "one_vs_rest"])
def test_class_seed_loop_over_a_labelled_set(ad_mode):
    """3 classes x 2 seeds through `ADTrainer.run` on a labelled resident image set: which rows are normal / anomalous is bit-exact
    with the reference's rule (oracle.batching, pinned by fixture g15), and the per-(class, seed) test scores and AUC equal the CPU
    oracle's run on the same splits and initial weights to 1e-3"""
    import eoe_amd
    from eoe_amd.data import LabelledImageSet
    from eoe_amd.models import ClipViTB32Custom
    from eoe_amd.training import TRAINER, ADTrainer
    eoe_amd.set_compute_dtype("fp16")
    n_cls, seeds, bs, epochs = 3, 2, 8, 2
    train, train_y, test, test_y, oe = _labelled_set(n_cls)
    mean, std = (0.5, 0.5, 0.5), (0.25, 0.25, 0.25)
    lset = LabelledImageSet(train, train_y, test, test_y, oe, classes=[f"c{k}" for k in range(n_cls)], crop=224, mean=mean, std=std,
                            noise_std=0.0)
    recorders = {}

    def dataset(c, seed):
        tr_ = trainer_ref[0]
        rec = _Recorder(lset.source(tr_.get_nominal_classes(c), seed))
        recorders[(c, seed)] = rec
        return rec
    presets = [[omodels.deterministic_init(ClipViTB32Custom(layers=1), tag=f"loo/{c}/{s}", layers=1) for s in range(seeds)]
               for c in range(n_cls)]
    tr = TRAINER["hsc"](ClipViTB32Custom(layers=1), dataset=dataset, epochs=epochs, lr=1e-4, wdk=1e-3, milestones=[1], batch_size=bs,
                        ad_mode=ad_mode, classes=lset.classes)
    trainer_ref = [tr]
    ADTrainer.KEEP_SNAPSHOT_IN_RAM = True
    try:
        models, res = tr.run(run_seeds=seeds, load=presets)
    finally:
        ADTrainer.KEEP_SNAPSHOT_IN_RAM = False
    assert len(res["cls_aucs"]) == n_cls and all(len(a) == seeds for a in res["cls_aucs"])
    worst_auc = worst_score = 0.0
    for c in range(n_cls):
        normal = batching.nominal_classes(ad_mode, c, n_cls)
        want_rows = batching.normal_subset(train_y.numpy(), normal)
        want_test = batching.ad_targets(test_y.numpy(), normal)
        for s in range(seeds):
            rec = recorders[(c, s)]
            # ---- membership, bit-exact: every epoch walks exactly the normal rows once; labels; OE offset; test labels
            per_epoch = len(rec.train_batches) // epochs
            for e in range(epochs):
                bt = rec.train_batches[e * per_epoch:(e + 1) * per_epoch]
                rows = np.concatenate([b[2][: len(b[2]) // 2].numpy() for b in bt])
                assert np.array_equal(np.sort(rows), want_rows), (ad_mode, c, s, e)
                for b in bt:
                    h = len(b[1]) // 2
                    assert b[1].tolist() == [0] * h + [1] * h
                    assert (b[2][h:] >= len(train_y)).all() and (b[2][h:] < len(train_y) + len(oe)).all()
            assert np.array_equal(torch.cat([b[1] for b in rec.test_batches]).numpy(), want_test)
            # ---- the oracle on the same splits and weights
            ref = omodels.deterministic_init(omodels.ClipViTNet(layers=1), tag=f"loo/{c}/{s}", layers=1)
            otrainer.train_steps(ref, [(b[0], b[1]) for b in rec.train_batches], "hsc", lr=1e-4, weight_decay=1e-3, milestones=[1],
                                 steps_per_epoch=per_epoch)
            ev = otrainer.eval_scores(ref, [(b[0], b[1]) for b in rec.test_batches], "hsc")
            with torch.no_grad():
                mm = models[c][s].cuda().eval()
                got = torch.cat([eoe_amd.hsc_score(mm(b[0].cuda())).cpu() for b in rec.test_batches]).numpy()
            worst_score = max(worst_score, float(np.abs(got - ev["scores"]).max()))
            n0, n1 = int((want_test == 0).sum()), int((want_test == 1).sum())
            d = abs(res["cls_aucs"][c][s] - ev["auc"])
            worst_auc = max(worst_auc, d)
            # 1e-3 on the AUC; with n0 x n1 test pairs one swapped near-tie moves it by 1 / (n0 n1), which is allowed for
            assert d <= max(1e-3, 1.0 / (n0 * n1) + 1e-12), (ad_mode, c, s, res["cls_aucs"][c][s], ev["auc"])
    print(f"[{ad_mode}] worst |score - oracle| {worst_score:.2e}, worst |AUC - oracle| {worst_auc:.2e}, mean AUC {res['mean_auc']:.4f}")
    assert worst_score < 1e-3
    assert 0.0 <= res["mean_auc"] <= 1.0


def test_batchnorm_encoders_train_in_exact_fp32_by_default():
    """`ADTrainer(exact_bn="auto")`: a BatchNorm encoder at lr >= 1e-3 is trained and scored with the exact-fp32 matrix-core convolutions
    (the mode that holds the trajectory bar), the ViT and small-lr runs keep the 16-bit path; the process-wide switch is restored"""
    import eoe_amd
    from eoe_amd import ops
    from eoe_amd.data import SyntheticAD
    from eoe_amd.models import CNN32
    from eoe_amd.training import TRAINER
    seen = []

    class Spy(TRAINER["hsc"]):
        def loss(self, *a, **k):
            seen.append(ops.parity_mode())
            return super().loss(*a, **k)

        def compute_anomaly_score(self, *a, **k):
            seen.append(ops.parity_mode())
            return super().compute_anomaly_score(*a, **k)
    ds = SyntheticAD(n_train_normal=16, n_oe=16, n_test=16, res=32, shift=1.0, seed=2)
    for lr, exact, want in ((1e-3, "auto", True), (1e-4, "auto", False), (1e-3, False, False), (1e-4, True, True)):
        seen.clear()
        tr = Spy(CNN32(bias=True), dataset=ds, epochs=1, lr=lr, wdk=0.0, milestones=[], batch_size=16, classes=["only"], exact_bn=exact)
        model, _ = tr.train_cls(copy.deepcopy(tr.model), ds, 0, "only", 0)
        tr.eval_cls(model, ds, 0, "only", 0)
        assert seen and all(s == want for s in seen), (lr, exact, seen)
        assert ops.parity_mode() is False


# --------------------------------------------------------------------------------------------- fp16-weights mode
def test_fp16_weights_sgd_matches_torch_on_half_tensors():
    """the reference's CLIP models carry fp16 parameters on a GPU (`convert_weights`, clip/model.py:371-392, applied by build_model :430)
    and are trained with torch.optim.SGD(momentum 0.9, nesterov) (ad_trainer.py:380-381).  FusedSGD on parameters marked by
    `eoe_amd.models.convert_weights` reproduces that update -- stock torch on half CUDA tensors is the oracle here: five steps, weight
    decay, fp16 gradients; the un-marked parameter of the same group keeps the fp32 update."""
    import eoe_amd
    from eoe_amd.optim import is_fp16_weight
    torch.manual_seed(3)
    lin = torch.nn.Linear(300, 72).cuda()
    ln = torch.nn.LayerNorm(72).cuda()
    from eoe_amd.models import convert_weights
    convert_weights(torch.nn.Sequential(lin, ln))
    assert is_fp16_weight(lin.weight) and is_fp16_weight(lin.bias) and not is_fp16_weight(ln.weight)
    assert torch.equal(lin.weight, lin.weight.half().float())
    ref = [torch.nn.Parameter(lin.weight.detach().half().clone()), torch.nn.Parameter(lin.bias.detach().half().clone()),
           torch.nn.Parameter(ln.weight.detach().clone())]
    kw = dict(lr=1e-2, momentum=0.9, nesterov=True, weight_decay=1e-3)
    opt_ref = torch.optim.SGD(ref, **kw)
    ours = [lin.weight, lin.bias, ln.weight]
    opt = eoe_amd.FusedSGD(ours, **kw)
    for step in range(5):
        for p, r in zip(ours, ref):
            g = torch.randn_like(r, dtype=torch.float32) * (0.5 + step)
            if r.dtype == torch.float16:
                g = g.half().float()                   # autograd hands an fp16 parameter an fp16 gradient
            p.grad, r.grad = g.clone(), g.to(r.dtype)
        opt.step()
        opt_ref.step()
    for p, r in zip(ours, ref):
        assert torch.isfinite(p).all()
        if r.dtype == torch.float16:
            assert torch.equal(p, p.half().float())                                  # the storage stays fp16-representable
            # the same fp32 arithmetic and the same two roundings per op (to fp32, then to fp16) as torch's multi-tensor kernels: equal bits.
            # (torch's own scalar tail path -- tensors whose length is not a multiple of 4 -- and its single-tensor path differ from its
            #  vectorised path in rare double-rounding ties, 0.01 .. 1 % of the elements by one fp16 ulp: tools/dbg_sgd16.py)
            assert torch.equal(p, r.detach().float()), ((p != r.detach().float()).float().mean().item(), (p - r.detach().float()).abs().max().item())
            mb = opt.state[p]["momentum_buffer"]
            assert torch.equal(mb, opt_ref.state[r]["momentum_buffer"].float())
        else:
            assert torch.allclose(p, r.detach(), rtol=1e-6, atol=1e-7)


def test_fp16_weights_mode_trains_the_vit_with_sgd():
    """end to end: a 2-layer ViT in fp16-weights mode takes SGD steps, its converted parameters stay fp16 values, LayerNorm / embeddings fp32"""
    import eoe_amd
    from eoe_amd.models import ClipViTB32Custom, convert_weights
    from eoe_amd.optim import is_fp16_weight
    torch.manual_seed(0)
    m = ClipViTB32Custom(layers=2).cuda().train()
    convert_weights(m.feature_model)                      # the CLIP tower only: the CustomNet head is created after build_model, in fp32
    marked = [n for n, p in m.named_parameters() if is_fp16_weight(p)]
    assert any("in_proj_weight" in n for n in marked) and any("c_fc.weight" in n for n in marked) and any(n.endswith("proj") for n in marked)
    assert not any("ln_" in n or "embedding" in n for n in marked)
    opt = eoe_amd.FusedSGD(m.parameters(), lr=1e-3, momentum=0.9, nesterov=True, weight_decay=1e-3)
    x = torch.randn(8, 3, 224, 224, device="cuda")
    y = torch.cat([torch.zeros(4, dtype=torch.long), torch.ones(4, dtype=torch.long)]).cuda()
    losses = []
    for _ in range(3):
        opt.zero_grad(set_to_none=True)
        loss = eoe_amd.hsc_loss(m(x), y, 0)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert all(np.isfinite(losses))
    for n, p in m.named_parameters():
        if is_fp16_weight(p):
            assert torch.equal(p, p.half().float()), n
    with pytest.raises(NotImplementedError):
        eoe_amd.FusedAdam(m.parameters(), lr=1e-4).step()


def test_vit_block_handover_equals_the_separate_pass():
    """the backward sweep hands dY of c_proj (the 16-bit copy of a block's dx_out) and its column sums from the producing block's
    LayerNorm-1 backward to the consuming block instead of recomputing them with eoe_cast_colsum: the same 16-bit values, so every
    gradient except the c_proj biases (column sums in another fixed order) keeps its bits"""
    import eoe_amd
    from eoe_amd import ops
    from eoe_amd.models import ClipViTB32Custom
    eoe_amd.set_compute_dtype("fp16")
    torch.manual_seed(0)
    m = ClipViTB32Custom(layers=3).cuda().train()
    x = torch.randn(16, 3, 224, 224, device="cuda")
    y = torch.cat([torch.zeros(8, dtype=torch.long), torch.ones(8, dtype=torch.long)]).cuda()
    res = {}
    old = ops.VIT_HANDOVER
    try:
        for on in (False, True, True):
            ops.VIT_HANDOVER = on
            for p in m.parameters():
                p.grad = None
            loss = eoe_amd.hsc_loss(m(x), y, 0)
            loss.backward()
            res.setdefault(on, []).append({n: p.grad.clone() for n, p in m.named_parameters()})
    finally:
        ops.VIT_HANDOVER = old
    off, on1, on2 = res[False][0], res[True][0], res[True][1]
    for n in off:
        assert torch.equal(on1[n], on2[n]), n                                   # reproducible
        if n.endswith("mlp.c_proj.bias") and "resblocks.2." not in n:           # the last block has no producer: its own pass
            d = (on1[n] - off[n]).abs().max().item() / (off[n].abs().max().item() + 1e-30)
            assert d < 1e-5, (n, d)
        else:
            assert torch.equal(on1[n], off[n]), n


def test_vit_last_block_on_class_token_rows_only_equals_the_full_block():
    """ln_post reads x[:, 0, :] only (clip/model.py:231-232): the last block run on its class-token rows alone (eoe_vit_block_fwd_args.cls_only)
    gives the same embedding -- the same kernels' per-row results -- and the gradients of the full block fed zeros on the other rows: equal
    up to the fp32 summation order of the last block's weight gradients (the zero rows are left out of the sums)"""
    import eoe_amd
    from eoe_amd import ops
    from eoe_amd.models import ClipViTB32Custom
    eoe_amd.set_compute_dtype("fp16")
    torch.manual_seed(0)
    m = ClipViTB32Custom(layers=3).cuda().train()
    y = torch.cat([torch.zeros(12, dtype=torch.long), torch.ones(12, dtype=torch.long)]).cuda()
    old = ops.VIT_CLS_ONLY_LAST
    # (the class-token-only block's n x 768 x 3072 GEMMs would otherwise run split over k -- another fp32 summation order, which the fp16
    #  rounding of the head's input turns into 1e-4 of the embedding: that form has its own test; here the two block forms run the same kernels)
    from eoe_amd import _lib
    _lib.check(_lib.lib.eoe_set_option(b"nt_flags", 8192), "eoe_set_option")
    try:
        for nimg in (24, 256):
            x = torch.randn(nimg, 3, 224, 224, device="cuda")
            yy = y if nimg == 24 else torch.cat([torch.zeros(128, dtype=torch.long), torch.ones(128, dtype=torch.long)]).cuda()
            res = {}
            for on in (False, True, True):
                ops.VIT_CLS_ONLY_LAST = on
                for p in m.parameters():
                    p.grad = None
                emb = m(x)
                loss = eoe_amd.hsc_loss(emb, yy, 0)
                loss.backward()
                torch.cuda.synchronize()
                res.setdefault(on, []).append((emb.detach().clone(), {n: p.grad.clone() for n, p in m.named_parameters()}))
            (e0, g0), (e1, g1), (e2, g2) = res[False][0], res[True][0], res[True][1]
            assert torch.equal(e1, e2)
            assert (e1 - e0).abs().max().item() <= 1e-6 * e0.abs().max().item(), (e1 - e0).abs().max().item()
            for n in g0:
                assert torch.equal(g1[n], g2[n]), n                                          # reproducible
                d = (g1[n] - g0[n]).abs().max().item() / (g0[n].abs().max().item() + 1e-30)
                assert d < 2e-5, (nimg, n, d)
        # forward only (scoring): the same embedding
        m.eval()
        with torch.no_grad():
            ops.VIT_CLS_ONLY_LAST = False
            a = m(x)
            ops.VIT_CLS_ONLY_LAST = True
            b = m(x)
        assert (a - b).abs().max().item() <= 1e-6 * a.abs().max().item()
    finally:
        ops.VIT_CLS_ONLY_LAST = old
        _lib.check(_lib.lib.eoe_set_option(b"nt_flags", 0), "eoe_set_option")


def test_vit_async_weight_gradients_keep_their_bits():
    """a block's grouped weight-gradient launch on the side stream, under the next block's kernels (ops.VIT_ASYNC_WGRAD): the same kernels
    on the same operands -- every gradient and a 6-step Adam trajectory are bitwise those of the synchronous launch order"""
    import copy
    import eoe_amd
    from eoe_amd import ops
    from eoe_amd.models import ClipViTB32Custom
    eoe_amd.set_compute_dtype("fp16")
    torch.manual_seed(0)
    m0 = ClipViTB32Custom(layers=4).cuda().train()
    x = torch.randn(32, 3, 224, 224, device="cuda")
    y = torch.cat([torch.zeros(16, dtype=torch.long), torch.ones(16, dtype=torch.long)]).cuda()
    out = {}
    old = ops.VIT_ASYNC_WGRAD
    try:
        for on in (False, True):
            ops.VIT_ASYNC_WGRAD = on
            m = copy.deepcopy(m0)
            opt = eoe_amd.FusedAdam(m.parameters(), lr=1e-4, weight_decay=1e-3)
            losses = []
            for _ in range(6):
                opt.zero_grad(set_to_none=True)
                loss = eoe_amd.hsc_loss(m(x), y, 0)
                loss.backward()
                if _ == 0:
                    g0 = {n: p.grad.clone() for n, p in m.named_parameters()}
                opt.step()
                losses.append(loss.item())
            out[on] = (losses, g0, {n: p.detach().clone() for n, p in m.named_parameters()})
    finally:
        ops.VIT_ASYNC_WGRAD = old
    (l0, g0, p0), (l1, g1, p1) = out[False], out[True]
    assert l0 == l1, (l0, l1)
    for n in g0:
        assert torch.equal(g0[n], g1[n]), n
        assert torch.equal(p0[n], p1[n]), n


def test_conv_async_weight_gradients_keep_their_bits():
    """the convolutions' weight-gradient GEMMs on a side stream (ops.CONV_ASYNC_WGRAD): a 4-step WideResNet-32 / CNN32 Adam trajectory is
    bitwise that of the synchronous order"""
    import copy
    import eoe_amd
    from eoe_amd import ops
    from eoe_amd.models import CNN32, WideResNet
    eoe_amd.set_compute_dtype("fp16")
    for make, res in ((lambda: CNN32(bias=True), 32), (lambda: WideResNet(res=32), 32)):
        torch.manual_seed(1)
        m0 = make().cuda().train()
        x = torch.randn(32, 3, res, res, device="cuda")
        y = torch.cat([torch.zeros(16, dtype=torch.long), torch.ones(16, dtype=torch.long)]).cuda()
        out = {}
        old = ops.CONV_ASYNC_WGRAD
        try:
            for on in (False, True):
                ops.CONV_ASYNC_WGRAD = on
                m = copy.deepcopy(m0)
                opt = eoe_amd.FusedAdam(m.parameters(), lr=1e-3)
                losses = []
                for _ in range(4):
                    opt.zero_grad(set_to_none=True)
                    loss = eoe_amd.hsc_loss(m(x), y, 0)
                    loss.backward()
                    opt.step()
                    losses.append(loss.item())
                out[on] = (losses, {n: p.detach().clone() for n, p in m.named_parameters()})
        finally:
            ops.CONV_ASYNC_WGRAD = old
        assert out[False][0] == out[True][0], (out[False][0], out[True][0])
        for n in out[False][1]:
            assert torch.equal(out[False][1][n], out[True][1][n]), n


@pytest.mark.parametrize("rows,cols", [(262144, 32), (4099, 64), (17, 128), (256, 512), (300, 2048), (33, 20)])
def test_f32_colsum(rows, cols):
    """the exact-fp32 mode's bias gradients: column sums of an fp32 matrix through the atomics-free partial-row scheme"""
    from eoe_amd import ops
    torch.manual_seed(rows + cols)
    x = torch.randn(rows, cols, device="cuda")
    out = torch.full((cols,), 7.0, device="cuda")
    ops._f32_colsum(x, out)
    want = x.double().sum(0)
    assert rel_rms(out, want.cpu()) < 2e-6, rel_rms(out, want.cpu())
    out2 = torch.full((cols,), 7.0, device="cuda")
    ops._f32_colsum(x, out2)
    assert torch.equal(out, out2)                       # fixed summation order


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_wide_two_workgroup_nt_kernel_keeps_the_bits(dtype):
    """the 160x256x32 two-workgroup NT kernel (default for N >= 2048 in multiples of 256 where its tiles fill the workgroup slots) adds the
    same products in the same k order as the kernels it replaces: plain, GELU-pair and GELU' x dY epilogues (with fused column sums), full
    and ragged M, bitwise equal to the launch with nt_flags bit 12 set"""
    from eoe_amd import ops, _lib

    def flags(v):
        _lib.check(_lib.lib.eoe_set_option(b"nt_flags", v), "eoe_set_option")

    g = torch.Generator(device="cuda").manual_seed(5)
    try:
        for m, n, k in ((12800, 3072, 768), (12763, 3072, 768), (6400, 2048, 256), (12800, 3072, 96)):
            a = (torch.randn(m, k, device="cuda", generator=g)).to(dtype)
            w = (torch.randn(n, k, device="cuda", generator=g) * 0.05).to(dtype)
            bias = torch.randn(n, device="cuda", generator=g)
            pre = torch.randn(m, n, device="cuda", generator=g).to(dtype)
            res = {}
            for f in (4096, 0):
                flags(f)
                o1 = torch.empty(m, n, device="cuda", dtype=dtype)
                ops.gemm_nt(a, w, o1, bias=bias)
                o2, p2 = torch.empty(m, n, device="cuda", dtype=dtype), torch.empty(m, n, device="cuda", dtype=dtype)
                ops.gemm_nt(a, w, o2, bias=bias, epilogue=ops.EPI_GELU, aux_out=p2)
                o3, cs = torch.empty(m, n, device="cuda", dtype=dtype), torch.zeros(n, device="cuda")
                ops.gemm_nt(a, w, o3, epilogue=ops.EPI_GELU_BWD, aux=pre, colsum_out=cs)
                o4 = torch.empty(m, n, device="cuda", dtype=torch.float32)
                ops.gemm_nt(a, w, o4, bias=bias, epilogue=ops.EPI_RESIDUAL, aux=pre.float())
                torch.cuda.synchronize()
                res[f] = (o1, o2, p2, o3, o4, cs)
            for x, y in zip(res[0][:5], res[4096][:5]):
                assert torch.equal(x, y), (m, n, k)
            d = (res[0][5] - res[4096][5]).abs().max().item() / (res[4096][5].abs().max().item() + 1e-30)
            assert d < 1e-5, (m, n, k, d)          # the column sums go through partial rows per 80 output rows either way
            ref = (a.float() @ w.float().t() + bias)
            assert (res[0][0].float() - ref).abs().max().item() < 0.05 * ref.abs().max().item()
    finally:
        flags(0)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_split_k_for_small_m_matches_the_plain_launch(dtype):
    """asked for with eoe_gemm_args.split_k, a small-M product behind K >= 1536 (the last ViT block on its class-token rows: 256 x 768 x 3072)
    runs as up to 8 k-ranges per tile + an in-order sum of the fp32 partial tiles: the plain launch's result to fp32 summation order,
    reproducible, independent of M row by row, with bias / fp32 residual / 16-bit output"""
    from eoe_amd import ops, _lib

    def flags(v):
        _lib.check(_lib.lib.eoe_set_option(b"nt_flags", v), "eoe_set_option")

    g = torch.Generator(device="cuda").manual_seed(11)
    try:
        for m, n, k in ((256, 768, 3072), (250, 768, 3072), (512, 256, 1536), (64, 3072, 2048)):
            a = torch.randn(m, k, device="cuda", generator=g).to(dtype)
            w = (torch.randn(n, k, device="cuda", generator=g) * 0.05).to(dtype)
            bias = torch.randn(n, device="cuda", generator=g)
            res = torch.randn(m, n, device="cuda", generator=g)
            out = {}
            for f in (8192, 0, 0):
                flags(f)
                o16 = torch.empty(m, n, device="cuda", dtype=dtype)
                ops.gemm_nt(a, w, o16, bias=bias, split_k=True)
                o32 = torch.empty(m, n, device="cuda", dtype=torch.float32)
                ops.gemm_nt(a, w, o32, bias=bias, epilogue=ops.EPI_RESIDUAL, aux=res, split_k=True)
                torch.cuda.synchronize()
                out.setdefault(f, []).append((o16, o32))
            (p16, p32), (s16, s32), (t16, t32) = out[8192][0], out[0][0], out[0][1]
            assert torch.equal(s16, t16) and torch.equal(s32, t32)                                   # reproducible
            if m > 8:                                                                                # a row's result does not depend on M
                h32 = torch.empty(8, n, device="cuda", dtype=torch.float32)
                ops.gemm_nt(a[:8].contiguous(), w, h32, bias=bias, epilogue=ops.EPI_RESIDUAL, aux=res[:8].contiguous(), split_k=True)
                assert torch.equal(h32, s32[:8]), (m, n, k)
            ref = a.float() @ w.float().t() + bias + res
            scale = ref.abs().max().item()
            assert (s32 - p32).abs().max().item() < 2e-6 * scale, (m, n, k)
            assert (s32 - ref).abs().max().item() < 1e-4 * scale, (m, n, k)
            assert (s16.float() - p16.float()).abs().max().item() <= 2.0 ** (-7 if dtype == torch.bfloat16 else -10) * scale
    finally:
        flags(0)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_adam_by_tiles_writes_the_operand_copies(dtype):
    """round 4 (`eoe_adam_tiles`): 2-D weights whose 16-bit operand copies exist are updated by 64 x 64 tiles and the copies rewritten in the
    same pass.  p / m / v are bit for bit the chunk kernel's, the copies bit for bit a fresh `cast_transpose` of the new weight, a skipped
    (non-finite) step leaves all of them alone, and the next `shadow.get` is a cache hit on the rewritten tensors."""
    import eoe_amd
    from eoe_amd import ops, optim
    eoe_amd.set_compute_dtype(dtype)
    shapes = ((768, 3072), (2304, 768), (100, 68), (64, 64), (33,), (4, 3, 2))          # ragged tiles, a vector, a 3-D tensor
    gen = torch.Generator(device="cuda").manual_seed(3)

    def make():
        return [torch.nn.Parameter(torch.randn(s, device="cuda", generator=torch.Generator(device="cuda").manual_seed(10 + i)) * 0.3)
                for i, s in enumerate(shapes)]
    a, b = make(), make()
    oa = eoe_amd.FusedAdam(a, lr=1e-2, weight_decay=1e-3, guard=True)
    ob = eoe_amd.FusedAdam(b, lr=1e-2, weight_decay=1e-3, guard=True)
    pairs = [ops.shadow.get(p, True, True) for p in a if p.dim() == 2]                     # the copies exist (as after a forward)
    ptrs = [(d.data_ptr(), t.data_ptr()) for d, t in pairs]
    for step in range(3):
        gs = [torch.randn(p.shape, device="cuda", generator=gen) * 0.1 for p in a]
        if step == 1:
            gs[0].view(-1)[12345] = float("inf")                                          # dropped whole by both paths
        for p, q, g in zip(a, b, gs):
            p.grad, q.grad = g.clone(), g.clone()
        assert optim.ADAM_TILES
        oa.step()
        optim.ADAM_TILES = False
        try:
            ob.step()
        finally:
            optim.ADAM_TILES = True
        torch.cuda.synchronize()
        for p, q in zip(a, b):
            assert torch.equal(p.detach(), q.detach())
            assert torch.equal(oa.state[p]["exp_avg"], ob.state[q]["exp_avg"]) and torch.equal(oa.state[p]["exp_avg_sq"], ob.state[q]["exp_avg_sq"])
        k = 0
        for p in a:
            if p.dim() != 2:
                continue
            d, t = ops.shadow.get(p, True, True)
            assert (d.data_ptr(), t.data_ptr()) == ptrs[k], "the rewritten copies are the cached ones"
            d2, t2 = ops.cast_transpose(p.detach(), dtype, True, True)
            assert torch.equal(d, d2) and torch.equal(t, t2)
            k += 1
    assert oa.skipped_steps() == 1 and ob.skipped_steps() == 1
