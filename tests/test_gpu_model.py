"""GPU tier 3: the drop-in modules (fused ViT block, whole CLIP ViT-B/32 CustomNet, fused Adam) against the
golden fixtures generated from the reference's own modules and against the CPU oracle on the same inputs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from gpu_util import DTYPES, EPS16, assert_close, rel_rms   # noqa: E402
from oracle import fill, models as omodels, trainer as otrainer, objectives   # noqa: E402

# stated tolerances (1e-3 on losses / scores).  fp16 operands (11-bit significand, what the
# reference's own GPU CLIP path uses, clip/model.py:371-392) meet the bar; bf16 (8-bit) is held to 4x that.
TRAJ_TOL = {torch.float16: 1e-3, torch.bfloat16: 4e-3}
BLOCK_RMS = {torch.float16: 1.5e-3, torch.bfloat16: 1.2e-2}


@pytest.fixture(autouse=True)
def _restore_dtype():
    import eoe_amd
    old = eoe_amd.compute_dtype()
    yield
    eoe_amd.set_compute_dtype(old)


@pytest.mark.parametrize("dtype", DTYPES)
def test_block_vs_golden(golden, dtype):
    import eoe_amd
    from eoe_amd.models import ResidualAttentionBlock
    eoe_amd.set_compute_dtype(dtype)
    g = golden("g4_block")
    blk = omodels.deterministic_init(ResidualAttentionBlock(768, 12), tag="blk", layers=12).cuda()
    x = torch.from_numpy(fill.fill("g4/x", (50, 2, 768), std=1.0)).permute(1, 0, 2).contiguous().cuda().requires_grad_(True)
    w = torch.from_numpy(fill.fill("g4/dy", (50, 2, 768), std=1.0)).permute(1, 0, 2).contiguous().cuda()
    y = blk(x)
    (y * w).sum().backward()
    yref = torch.from_numpy(g["y"]).permute(1, 0, 2)
    dxref = torch.from_numpy(g["dx"]).permute(1, 0, 2)
    ry, rdx = rel_rms(y, yref), rel_rms(x.grad, dxref)
    print(f"[block {dtype}] rel rms: y {ry:.2e} dx {rdx:.2e}")
    assert ry < BLOCK_RMS[dtype] and rdx < BLOCK_RMS[dtype], (ry, rdx)
    for n, p in blk.named_parameters():
        ref = float(g[f"gnorm/{n}"])
        got = p.grad.double().norm().item()
        assert abs(got - ref) <= 2 * BLOCK_RMS[dtype] * ref + 1e-6, (n, got, ref)
        head = torch.from_numpy(g[f"ghead/{n}"]).double()
        err = (p.grad.flatten()[:16].cpu().double() - head).abs().max().item()
        assert err <= 8 * BLOCK_RMS[dtype] * max(head.abs().max().item(), ref / np.sqrt(p.numel())), (n, err)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("layers,n_half,steps,obj,freeze", [
    (2, 2, 3, "hsc", False), (2, 2, 3, "hsc", True), (12, 1, 2, "hsc", False), (2, 2, 3, "bce", False)])
def test_vit_trajectory_vs_golden(golden, dtype, layers, n_half, steps, obj, freeze):
    import eoe_amd
    from eoe_amd.models import ClipViTB32Custom
    eoe_amd.set_compute_dtype(dtype)
    tag = f"g3_vit_l{layers}_{obj}" + ("_frozen" if freeze else "")
    g = golden(tag)
    m = ClipViTB32Custom(clf=(obj == "bce"), freeze=freeze, layers=layers)
    omodels.deterministic_init(m, tag="vit", layers=layers)
    m = m.cuda().train()
    opt = eoe_amd.FusedAdam(m.parameters(), lr=1e-4, weight_decay=1e-3)
    if freeze:
        m.freeze_parts()
    batches = [otrainer.synthetic_batch(f"g3/b{i}", n_half, n_half, 224) for i in range(steps)]
    losses, scores = [], []
    for it, (imgs, lbls) in enumerate(batches):
        imgs, lbls = imgs.cuda(), lbls.cuda()
        opt.zero_grad()
        feats = m(imgs)
        loss = eoe_amd.hsc_loss(feats, lbls, 0) if obj == "hsc" else eoe_amd.bce_loss(feats, lbls)
        loss.backward()
        if it == 0:
            f0 = feats.detach().clone()
            grads0 = {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
        opt.step()
        opt.zero_grad()
        losses.append(loss.item())
        scores.append((eoe_amd.hsc_score(feats) if obj == "hsc" else eoe_amd.bce_score(feats)).cpu().numpy())
    tol = TRAJ_TOL[dtype]
    rf = rel_rms(f0, torch.from_numpy(g["features0"]))
    dl = np.abs(np.array(losses) - g["losses"]) / np.maximum(1.0, np.abs(g["losses"]))
    ds = np.abs(np.stack(scores) - g["scores"]).max()
    print(f"[vit l{layers} frozen={freeze} {dtype}] features rel rms {rf:.2e}; loss dev {dl.max():.2e}; score dev {ds:.2e}")
    print("   losses", losses, "golden", g["losses"].tolist())
    assert dl.max() <= tol, (losses, g["losses"])
    assert ds <= tol, ds
    # first-step gradients: per-tensor norms against the reference's
    worst = 0.0
    for n, gr in grads0.items():
        ref = float(g[f"gnorm/{n}"])
        dev = abs(gr.double().norm().item() - ref) / max(ref, 1e-12)
        worst = max(worst, dev)
        assert dev <= 30 * EPS16[dtype] + 1e-3, (n, dev, ref)
    print(f"   worst grad-norm deviation {worst:.2e}")
    if freeze:
        assert all(n.startswith("final_linear") for n in grads0)
    # encoder output after the K steps (parameters moved the same way)
    with torch.no_grad():
        m.eval()
        enc = m.feature_model(batches[0][0].cuda())
    re = rel_rms(enc, torch.from_numpy(g["enc_after"]))
    print(f"   encoder-after rel rms {re:.2e}")
    assert re < 20 * EPS16[dtype], re


def test_state_dict_interchange():
    """a state_dict of the oracle (= the reference's names and shapes) loads unchanged, and back"""
    from eoe_amd.models import ClipViTB32Custom
    ref = omodels.ClipViTNet(layers=1)
    omodels.deterministic_init(ref, tag="sd", layers=1)
    m = ClipViTB32Custom(layers=1)
    missing = m.load_state_dict(ref.state_dict(), strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    ref2 = omodels.ClipViTNet(layers=1)
    ref2.load_state_dict(m.state_dict(), strict=True)
    m = m.cuda()
    x = torch.from_numpy(fill.fill("sd/x", (2, 3, 224, 224), std=1.0))
    with torch.no_grad():
        got = m(x.cuda())
        want = ref(x)
    assert rel_rms(got, want) < 2e-2


def test_vit_step_is_bitwise_reproducible():
    """three runs of 4 Adam steps of a 2-layer ViT-B/32 from the same state give identical bits: every reduction of the step
    (LayerNorm parameter / bias gradients, wgrad split sums, loss) goes through partial rows and fixed-order finish kernels -- an
    fp32 atomic or a race anywhere would show up here"""
    import copy
    import eoe_amd
    from eoe_amd.models import ClipViTB32Custom
    eoe_amd.set_compute_dtype("fp16")
    torch.manual_seed(5)
    m0 = ClipViTB32Custom(prediction_head=True, clf=False, layers=2).cuda().train()
    x, y = otrainer.synthetic_batch("det/vit", 12, 12, 224)
    x, y = x.cuda(), y.cuda()
    runs = []
    for _ in range(3):
        m = copy.deepcopy(m0)
        opt = eoe_amd.FusedAdam(m.parameters(), lr=1e-4, weight_decay=1e-3)
        losses = []
        for _ in range(4):
            opt.zero_grad()
            loss = eoe_amd.hsc_loss(m(x), y, 0)
            loss.backward()
            opt.step()
            losses.append(loss.item())
        runs.append((losses, {k: v.detach().clone() for k, v in m.state_dict().items()}))
    for losses, sd in runs[1:]:
        assert losses == runs[0][0]
        for k in sd:
            assert torch.equal(sd[k], runs[0][1][k]), k
