"""GPU tier: the WideResNet + CBAM path -- general conv-as-GEMM (3x3 s1/s2, 1x1 s2, 7x7 s2 stem),
MaxPool 3x3/2, CBAM gates, residual add+ReLU, global average pool and the drop-in module, against the oracle and the
golden vectors generated from the reference's own WideResNet / CBAM (tests/golden/g5_*)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from gpu_util import DTYPES, EPS16, f32, assert_close, rel_rms, conditioned_tol   # noqa: E402
from oracle import fill, models as omodels, objectives, trainer as otrainer   # noqa: E402


@pytest.fixture(autouse=True)
def _restore_dtype():
    import eoe_amd
    old = eoe_amd.compute_dtype()
    yield
    eoe_amd.set_compute_dtype(old)


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("n,cin,cout,H,k,s,p,slope,is_image", [
    (2, 64, 64, 14, 3, 1, 1, 0.0, False),     # BasicBlock conv1 (resnet.py:133-135)
    (3, 64, 128, 14, 3, 2, 1, 0.0, False),    # stride-2 conv1 of layer2..4
    (2, 128, 128, 7, 3, 1, 1, 1.0, False),    # conv2 + bn2 without activation
    (2, 64, 128, 14, 1, 2, 0, 1.0, False),    # downsample 1x1 stride 2 (resnet.py:72-76)
    (2, 3, 64, 32, 7, 2, 3, 0.0, True),       # the stem (resnet.py:35,93-95) on a small image
    (2, 8, 64, 9, 3, 2, 1, 0.0, False),       # odd grid, narrow input (dgrad GEMM needs cout % 64 == 0)
])
def test_conv_bn_general_vs_oracle(dtype, n, cin, cout, H, k, s, p, slope, is_image):
    import eoe_amd
    import eoe_amd.ops as ops
    eoe_amd.set_compute_dtype(dtype)
    x, xr = f32("rc/x", (n, cin, H, H), 1.0)
    w, wr = f32("rc/w", (cout, cin, k, k), (1.0 / (k * k * cin)) ** 0.5)
    g, gr = f32("rc/g", (cout,), 0.1, mean=1.0)
    b, br = f32("rc/b", (cout,), 0.1)
    rm, rv = torch.zeros(cout, device="cuda"), torch.ones(cout, device="cuda")
    nbt = torch.zeros((), dtype=torch.long, device="cuda")
    mean = torch.tensor([0.1, -0.2, 0.05], device="cuda") if is_image else None
    std = torch.tensor([0.9, 1.1, 1.3], device="cuda") if is_image else None
    xin = (x if is_image else _nhwc(x)).requires_grad_(not is_image)
    wg, gg, bg = (t.clone().requires_grad_(True) for t in (w, g, b))
    cfg = (True, 1e-5, 0.1, 1, is_image, mean, std, False, (k, k, s, p), slope)
    out = ops.conv_bn_act_pool(xin, wg, None, gg, bg, rm, rv, nbt, cfg)
    xd = xr.double()
    if is_image:
        xd = (xd - mean.cpu().double().view(1, 3, 1, 1)) / std.cpu().double().view(1, 3, 1, 1)
    xd = xd.to(dtype).double().requires_grad_(True)
    wd = wr.to(dtype).double().requires_grad_(True)
    gd, bd = (t.double().requires_grad_(True) for t in (gr, br))
    rmr, rvr = torch.zeros(cout, dtype=torch.float64), torch.ones(cout, dtype=torch.float64)
    zr = omodels.batch_norm(F.conv2d(xd, wd, None, stride=s, padding=p), gd, bd, rmr, rvr, True, 0.1, 1e-5)
    want = F.leaky_relu(zr, slope).permute(0, 2, 3, 1)
    assert_close(out, want, 1e-3, 2e-3, "conv+bn forward")
    assert_close(rm, rmr, 1e-3, 1e-4, "running_mean")
    assert_close(rv, rvr, 1e-3, 1e-4, "running_var")
    dout, doutr = f32("rc/dout", tuple(out.shape), 1.0)
    (out * dout).sum().backward()
    (want * doutr.double()).sum().backward()
    tol = 30 * EPS16[dtype]
    for name, got, ref in (("dw", wg.grad, wd.grad), ("dgamma", gg.grad, gd.grad), ("dbeta", bg.grad, bd.grad)):
        r = rel_rms(got, ref)
        assert r < tol, (name, r)
    if not is_image:
        r = rel_rms(xin.grad, xd.grad.permute(0, 2, 3, 1))
        assert r < tol, ("dx", r)


@pytest.mark.parametrize("n,cin,cout,H,k,s,p", [(5, 64, 64, 13, 3, 1, 1), (3, 128, 256, 14, 3, 2, 1), (2, 64, 128, 9, 1, 2, 0),
                                               (2, 256, 64, 7, 3, 1, 1), (1, 64, 192, 30, 5, 1, 2), (3, 32, 64, 16, 5, 1, 2),
                                               (2, 16, 64, 11, 3, 1, 1), (2, 8, 64, 10, 3, 2, 1),
                                               # 1x1 maps under a padded kernel: the one-tap plain-GEMM shortcut (ops._single_pixel)
                                               (37, 128, 192, 1, 3, 1, 1), (70, 64, 64, 1, 5, 1, 2), (256, 512, 512, 1, 3, 1, 1)])
def test_implicit_conv_equals_materialised(n, cin, cout, H, k, s, p):
    """the implicit-GEMM path (patches fetched inside the GEMM's LDS stage: forward, wgrad, stride-1 dgrad) against the
    materialised im2col / col2im path on the same 16-bit operands: same products, different summation order only"""
    import eoe_amd.ops as ops
    x, _ = f32("ic/x", (n, H, H, cin), 1.0)
    w, _ = f32("ic/w", (cout, cin, k, k), (1.0 / (k * k * cin)) ** 0.5)
    g, _ = f32("ic/g", (cout,), 0.1, mean=1.0)
    b, _ = f32("ic/b", (cout,), 0.1)
    res = {}
    for mode in (True, False):
        ops.set_implicit_conv(mode)
        try:
            xin = x.clone().requires_grad_(True)
            wg, gg, bg = (t.clone().requires_grad_(True) for t in (w, g, b))
            rm, rv = torch.zeros(cout, device="cuda"), torch.ones(cout, device="cuda")
            nbt = torch.zeros((), dtype=torch.long, device="cuda")
            cfg = (True, 1e-5, 0.1, 1, False, None, None, False, (k, k, s, p), 0.0)
            out = ops.conv_bn_act_pool(xin, wg, None, gg, bg, rm, rv, nbt, cfg)
            dout, _ = f32("ic/dout", tuple(out.shape), 1.0)
            (out * dout).sum().backward()
            res[mode] = (out.detach(), xin.grad, wg.grad, gg.grad, bg.grad)
        finally:
            ops.set_implicit_conv(True)
    for name, a, bb in zip(("out", "dx", "dw", "dgamma", "dbeta"), res[True], res[False]):
        r = rel_rms(a, bb.cpu())
        assert r < 2e-3, (name, r)        # dy16 is rounded to 16 bit after slightly different fp32 sums: a few flipped roundings


@pytest.mark.parametrize("n,C,H,W", [(2, 64, 16, 16), (3, 16, 9, 7), (1, 64, 112, 112)])
def test_maxpool_3x3_s2(n, C, H, W):
    """nn.MaxPool2d(3, 2, 1) incl. ties after a ReLU (the first maximum takes the gradient) -- exact"""
    import eoe_amd.ops_resnet as R
    x, xr = f32("mp/x", (n, C, H, W), 1.0)
    x, xr = torch.relu(x), torch.relu(xr)                       # about half the entries tie at 0
    xin = _nhwc(x).requires_grad_(True)
    out = R.max_pool(xin, 3, 2, 1)
    xd = xr.clone().requires_grad_(True)
    want = F.max_pool2d(xd, 3, 2, 1)
    assert torch.equal(out.detach().cpu(), want.detach().permute(0, 2, 3, 1))
    dout, doutr = f32("mp/dout", tuple(want.shape), 1.0)
    (out * _nhwc(dout)).sum().backward()
    (want * doutr).sum().backward()
    assert_close(xin.grad, xd.grad.permute(0, 2, 3, 1), 1e-6, 1e-6, "maxpool dx")


def _load_cbam(mod, ref):
    mod.load_state_dict(ref.state_dict())
    return mod.cuda()


def test_cbam_vs_golden(golden):
    """CBAM(64) forward, input gradient, every parameter gradient and the spatial BatchNorm buffers against the vectors
    produced by the reference's own cbam.py (fp32 kernels: tight tolerances)"""
    from eoe_amd.models import CBAM
    g = golden("g5_cbam")
    ref = omodels.deterministic_init(omodels._CBAM(64), tag="cbam")
    cb = _load_cbam(CBAM(64), ref)
    cb.train()
    x = torch.from_numpy(fill.fill("g5/cbam_x", (2, 64, 14, 14), std=1.0))
    w = torch.from_numpy(fill.fill("g5/cbam_dy", (2, 64, 14, 14), std=1.0))
    xin = _nhwc(x).cuda().requires_grad_(True)
    y = cb(xin)
    (y * _nhwc(w).cuda()).sum().backward()
    assert_close(y, torch.from_numpy(g["y"]).permute(0, 2, 3, 1), 1e-4, 1e-5, "cbam y")
    assert_close(xin.grad, torch.from_numpy(g["dx"]).permute(0, 2, 3, 1), 1e-3, 1e-4, "cbam dx")
    for name, p in cb.named_parameters():
        gn = p.grad.double().norm().item()
        assert abs(gn - float(g[f"gnorm/{name}"])) <= 1e-3 * float(g[f"gnorm/{name}"]) + 1e-6, name
        head = p.grad.detach().reshape(-1)[: g[f"ghead/{name}"].shape[0]].cpu().numpy()
        np.testing.assert_allclose(head, g[f"ghead/{name}"], rtol=2e-3, atol=1e-5, err_msg=name)
    for name, b in cb.named_buffers():
        np.testing.assert_allclose(b.cpu().numpy(), g[f"buf/{name}"], rtol=1e-4, atol=1e-6, err_msg=name)


@pytest.mark.parametrize("n,C,H", [(3, 128, 7), (2, 512, 7), (5, 256, 5)])
def test_cbam_vs_oracle_shapes(n, C, H):
    """other widths / odd grids, train and eval mode, against the oracle module in fp64"""
    from eoe_amd.models import CBAM
    ref = omodels.deterministic_init(omodels._CBAM(C), tag=f"cbam{C}")
    with torch.no_grad():
        ref.SpatialGate.spatial.bn.weight.fill_(0.7)
        ref.SpatialGate.spatial.bn.bias.fill_(-0.1)
    cb = _load_cbam(CBAM(C), ref)
    refd = ref.double()
    for training in (True, False):
        cb.train(training)
        refd.train(training)
        x = torch.from_numpy(fill.fill(f"cb/x{C}", (n, C, H, H), std=1.0))
        w = torch.from_numpy(fill.fill(f"cb/dy{C}", (n, C, H, H), std=1.0))
        xin = _nhwc(x).cuda().requires_grad_(True)
        xd = x.double().requires_grad_(True)
        cb.zero_grad()
        refd.zero_grad()
        y = cb(xin)
        yr = refd(xd)
        (y * _nhwc(w).cuda()).sum().backward()
        (yr * w.double()).sum().backward()
        assert_close(y, yr.permute(0, 2, 3, 1), 1e-4, 1e-5, f"y train={training}")
        assert_close(xin.grad, xd.grad.permute(0, 2, 3, 1), 1e-3, 1e-4, f"dx train={training}")
        for (name, p), (_, pr) in zip(cb.named_parameters(), refd.named_parameters()):
            r = rel_rms(p.grad, pr.grad)
            assert r < 2e-4, (name, training, r)


def test_add_relu_and_avgpool():
    import eoe_amd.ops_resnet as R
    a, ar = f32("ar/a", (2, 7, 7, 512), 1.0)
    b, br = f32("ar/b", (2, 7, 7, 512), 1.0)
    a.requires_grad_(True), b.requires_grad_(True)
    ad, bd = ar.clone().requires_grad_(True), br.clone().requires_grad_(True)
    out = R.GlobalAvgPoolFunction.apply(R.add_relu(a, b))
    want = torch.relu(ad + bd).mean(dim=(1, 2))
    assert_close(out, want, 1e-6, 1e-6, "avgpool(relu(a+b))")
    dout, doutr = f32("ar/d", (2, 512), 1.0)
    (out * dout).sum().backward()
    (want * doutr).sum().backward()
    assert_close(a.grad, ad.grad, 1e-6, 1e-7, "da")
    assert_close(b.grad, bd.grad, 1e-6, 1e-7, "db")


@pytest.mark.parametrize("y16", [False, True])
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cin,planes,stride", [(64, 64, 1), (64, 128, 2)])
def test_basic_block_vs_oracle(dtype, cin, planes, stride, y16, monkeypatch):
    """one BasicBlock (conv-bn-relu-conv-bn [+ downsample] -> CBAM -> +res -> relu) forward + all gradients"""
    import eoe_amd
    from eoe_amd.models.resnet import BasicBlock
    import torch.nn as nn
    eoe_amd.set_compute_dtype(dtype)
    ds = stride != 1 or cin != planes
    ref = omodels.deterministic_init(omodels.BasicBlock(cin, planes, stride, ds), tag="bb")
    down = nn.Sequential(nn.Conv2d(cin, planes, 1, stride=stride, bias=False), nn.BatchNorm2d(planes)) if ds else None
    blk = BasicBlock(cin, planes, stride, down, use_cbam=True)
    blk.load_state_dict(ref.state_dict())
    blk = blk.cuda().train()
    x = torch.from_numpy(fill.fill("bb/x", (4, cin, 14, 14), std=1.0))
    w = torch.from_numpy(fill.fill("bb/dy", (4, planes, 14 // stride, 14 // stride), std=1.0))
    from eoe_amd import ops as _ops
    if y16 and dtype != torch.float16:
        pytest.skip("the fp16 convolution output (ops.CONV_Y16) is an fp16-only option")
    monkeypatch.setattr(_ops, "CONV_Y16", y16)
    if _ops.CONV_Y16 and dtype == torch.float16:
        # the kernels keep a convolution's output in fp16 between the GEMM and BatchNorm (ops.CONV_Y16, channel counts that are multiples
        # of 16): the fp64 reference sees the same rounded values (straight-through for the gradient), so that the ReLU masks agree
        class _RoundedConv:
            def __getattr__(self, k):
                return getattr(F, k)

            @staticmethod
            def conv2d(inp, weight, bias=None, **kw):
                o = F.conv2d(inp, weight, bias, **kw)
                return o + (o.to(dtype).to(o.dtype) - o).detach() if weight.shape[0] % 16 == 0 else o
        monkeypatch.setattr(omodels, "F", _RoundedConv())
    # how far the oracle's own fp32 evaluation is from fp64, per tensor (conditioning of the 4-image BatchNorm statistics)
    x32 = x.clone().requires_grad_(True)
    ref.train()
    (ref(x32) * w).sum().backward()
    g32 = {n: p.grad.clone() for n, p in ref.named_parameters()}
    ref.zero_grad()
    ref.load_state_dict(blk.state_dict())            # undo the running-statistics update of that pass
    refd = ref.double().train()
    xin = _nhwc(x).cuda().requires_grad_(True)
    xd = x.double().requires_grad_(True)
    y = blk(xin)
    yr = refd(xd)
    (y * _nhwc(w).cuda()).sum().backward()
    (yr * w.double()).sum().backward()
    tol = 40 * EPS16[dtype]
    assert rel_rms(y, yr.permute(0, 2, 3, 1)) < tol
    assert rel_rms(xin.grad, xd.grad.permute(0, 2, 3, 1)) < tol
    bad, pinned = {}, 0
    for (name, p), (_, pr) in zip(blk.named_parameters(), refd.named_parameters()):
        if pr.grad.abs().max().item() < 1e-9 or "SpatialGate.spatial.bn" in name:
            continue      # the two scalar gate-BN gradients are cancelling sums over all pixels: pinned exactly (fp32 kernels,
            #               exact inputs) by test_cbam_vs_golden / test_cbam_vs_oracle_shapes, noise behind a 16-bit conv
        t = conditioned_tol(2 * tol, rel_rms(g32[name], pr.grad), dtype)
        if t is None:
            continue                                  # cancellation noise at this size (the scalar spatial-gate BN gradients)
        pinned += 1
        r = rel_rms(p.grad, pr.grad)
        if r > t:
            bad[name] = (r, t)
    assert not bad and pinned >= 10, (bad, pinned)
    for (name, bf), (_, bfr) in zip(blk.named_buffers(), refd.named_buffers()):
        assert_close(bf, bfr, 2e-3, 4 * EPS16[dtype], name)


def test_wideresnet_state_dict_and_param_count():
    from eoe_amd.models import WideResNet
    m = WideResNet()
    o = omodels.WideResNet()
    assert [(k, tuple(v.shape)) for k, v in m.state_dict().items()] == [(k, tuple(v.shape)) for k, v in o.state_dict().items()]
    assert sum(p.numel() for p in m.parameters()) == 11397720
    # initialisation rule of resnet.py:54-66
    sd = m.state_dict()
    assert float(sd["layer1.0.cbam.SpatialGate.spatial.bn.weight"].abs().max()) == 0.0
    assert float(sd["layer3.1.bn2.weight"].min()) == 1.0 and float(sd["fc.bias"].abs().max()) == 0.0


@pytest.mark.parametrize("parity", [False, True])
def test_wideresnet_small_batch_vs_golden(golden, parity):
    """WideResNet@224 at N = 4 (2 normal + 2 OE) against the vectors of the reference's own resnet.py (fixture g5): features and
    the first loss.  BatchNorm over 4 images is ill-conditioned by construction (the reference's own fp32 run is 1.2e-2 from
    its fp64 run on the worst gradient tensor), so gradients and K-step trajectories are pinned on the well-conditioned
    fixtures instead (tests/test_gpu_parity_big.py: 16 + 16 images, 10 steps, 1e-3 in parity mode)."""
    import eoe_amd
    from eoe_amd.models import WideResNet
    from eoe_amd.ops import hsc_loss
    eoe_amd.set_compute_dtype(torch.float16)
    eoe_amd.set_parity_mode(parity)
    try:
        g = golden("g5_wideresnet_hsc")
        ref = omodels.deterministic_init(omodels.WideResNet(), tag="wrn")
        m = WideResNet()
        m.load_state_dict(ref.state_dict())
        m = m.cuda().train()
        x0, y0 = (t.cuda() for t in otrainer.synthetic_batch("g5/b0", 2, 2, 224))
        f0 = m(x0)
        r = rel_rms(f0, torch.from_numpy(g["features0"]))
        assert r < (1e-3 if parity else 2e-2), r
        loss = hsc_loss(f0, y0, 0)
        loss.backward()
        assert abs(loss.item() - float(g["losses"][0])) <= 1e-4 * abs(float(g["losses"][0]))
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
    finally:
        eoe_amd.set_parity_mode(False)


def test_wideresnet_eval_mode_and_clf():
    """eval mode uses the running statistics everywhere (ad_trainer.py:480 model.eval()); clf head -> N x 1"""
    from eoe_amd.models import WideResNet
    ref = omodels.deterministic_init(omodels.WideResNet(clf=True), tag="wrnc")
    m = WideResNet(clf=True)
    m.load_state_dict(ref.state_dict())
    m = m.cuda().eval()
    ref = ref.eval()
    x = otrainer.synthetic_batch("g5/e", 1, 1, 224)[0]
    with torch.no_grad():
        out = m(x.cuda())
        want = ref(x)
    assert out.shape == (2, 1)
    assert rel_rms(out, want) < 2e-2


def test_wideresnet_full_batch_properties():
    """at the benchmark's full step batch (128 + 128 images of 224x224): fp16 and bf16 paths agree per tensor (no fp16
    gradient underflow; BatchNorm over 256 images is well conditioned), the implicit-GEMM path equals the materialised one at
    the full size, the loss equals the mean of the per-sample losses recomputed from the returned features, and one
    optimiser step moves every parameter that has a non-zero gradient."""
    import eoe_amd
    import eoe_amd.ops as ops
    from eoe_amd.models import WideResNet
    torch.manual_seed(0)
    n = 256
    x = torch.randn(n, 3, 224, 224, device="cuda")
    x[n // 2:] += 0.5 * torch.randn(1, 3, 224, 224, device="cuda")
    y = torch.cat([torch.zeros(n // 2, dtype=torch.long), torch.ones(n // 2, dtype=torch.long)]).cuda()
    m = WideResNet().cuda().train()
    with torch.no_grad():                        # the reference's zero init of the gate BN weight hides the spatial gate
        for k, p in m.named_parameters():
            if k.endswith("SpatialGate.spatial.bn.weight"):
                p.fill_(0.5)
    res = {}
    for tag, dt, implicit in (("fp16", "fp16", True), ("bf16", "bf16", True), ("fp16m", "fp16", False)):
        eoe_amd.set_compute_dtype(dt)
        ops.set_implicit_conv(implicit)
        try:
            for p in m.parameters():
                p.grad = None
            f = m(x)
            loss = eoe_amd.hsc_loss(f, y, 0)
            loss.backward()
            res[tag] = (loss.item(), f.detach().clone(), {k: p.grad.double().norm().item() for k, p in m.named_parameters()})
        finally:
            ops.set_implicit_conv(True)
    l16, f16, g16 = res["fp16"]
    lb, fb, gb = res["bf16"]
    lm, fm, gm = res["fp16m"]
    assert np.isfinite(l16) and abs(l16 - lb) < 1e-2 * max(1, abs(l16)), (l16, lb)
    d_b = sorted(abs(g16[k] - gb[k]) / max(gb[k], 1e-12) for k in g16)
    d_m = sorted(abs(g16[k] - gm[k]) / max(gm[k], 1e-12) for k in g16)
    print(f"[wrn full batch] loss fp16 {l16:.6f} bf16 {lb:.6f} materialised {lm:.6f}; grad-norm rel diff fp16 vs bf16 median "
          f"{d_b[len(d_b) // 2]:.2e} max {d_b[-1]:.2e}; implicit vs materialised median {d_m[len(d_m) // 2]:.2e} max {d_m[-1]:.2e}")
    assert d_b[len(d_b) // 2] < 2e-2 and d_b[int(0.9 * len(d_b))] < 1e-1
    assert abs(l16 - lm) < 1e-3 * max(1, abs(l16)) and rel_rms(f16, fm.cpu()) < 5e-3
    assert d_m[len(d_m) // 2] < 5e-3 and d_m[int(0.9 * len(d_m))] < 5e-2
    want = objectives.hsc_loss(f16.double().cpu(), y.cpu()).item()
    assert abs(l16 - want) < 1e-5 * max(1, abs(want))
    before = {k: p.detach().clone() for k, p in m.named_parameters()}
    opt = eoe_amd.FusedAdam(m.parameters(), lr=1e-3, weight_decay=0.0)
    opt.step()
    still = [k for k, p in m.named_parameters() if gm[k] > 0 and torch.equal(before[k], p.detach())]
    assert not still, still


@pytest.mark.parametrize("y16", [False, True])
@pytest.mark.parametrize("dtype", DTYPES)
def test_stem_conv_bn_relu_maxpool_fused(dtype, y16, monkeypatch):
    """conv7x7/2 -> BN -> ReLU -> MaxPool(3,2,1) as ONE unit (packed first layer + fused BN/ReLU/overlapping max-pool kernels)
    against torch-CPU fp64, forward and every gradient"""
    import eoe_amd
    import eoe_amd.ops as ops
    if y16 and dtype != torch.float16:
        pytest.skip("the fp16 convolution output (ops.CONV_Y16) is an fp16-only option")
    monkeypatch.setattr(ops, "CONV_Y16", y16)
    eoe_amd.set_compute_dtype(dtype)
    n, cout, H = 3, 64, 36
    x, xr = f32("st/x", (n, 3, H, H), 1.0)
    w, wr = f32("st/w", (cout, 3, 7, 7), (1.0 / 147) ** 0.5)
    g, gr = f32("st/g", (cout,), 0.1, mean=1.0)
    b, br = f32("st/b", (cout,), 0.1)
    rm, rv = torch.zeros(cout, device="cuda"), torch.ones(cout, device="cuda")
    nbt = torch.zeros((), dtype=torch.long, device="cuda")
    mean = torch.tensor([0.1, -0.2, 0.05], device="cuda")
    std = torch.tensor([0.9, 1.1, 1.3], device="cuda")
    wg, gg, bg = (t.clone().requires_grad_(True) for t in (w, g, b))
    cfg = (True, 1e-5, 0.1, (3, 2, 1), True, mean, std, False, (7, 7, 2, 3), 0.0, True)
    out = ops.conv_bn_act_pool(x, wg, None, gg, bg, rm, rv, nbt, cfg)
    assert out._eoe16.dtype == dtype and rel_rms(out._eoe16.float(), out.detach().cpu()) < 2 * EPS16[dtype]
    xd = ((xr.double() - mean.cpu().double().view(1, 3, 1, 1)) / std.cpu().double().view(1, 3, 1, 1)).to(dtype).double()
    wd = wr.to(dtype).double().requires_grad_(True)
    gd, bd = (t.double().requires_grad_(True) for t in (gr, br))
    rmr, rvr = torch.zeros(cout, dtype=torch.float64), torch.ones(cout, dtype=torch.float64)
    yr = F.conv2d(xd, wd, None, stride=2, padding=3)
    if ops.CONV_Y16 and dtype == torch.float16:
        # the kernels keep the convolution output in fp16 between the GEMM and BatchNorm (statistics from the fp32 accumulators): the
        # reference normalises the same rounded values with the unrounded statistics (straight-through for the gradient)
        yq = yr + (yr.to(dtype).double() - yr).detach()
        mu, var = yr.mean((0, 2, 3), keepdim=True), yr.var((0, 2, 3), unbiased=False, keepdim=True)
        zr = torch.relu((yq - mu) / torch.sqrt(var + 1e-5) * gd.view(1, -1, 1, 1) + bd.view(1, -1, 1, 1))
    else:
        zr = torch.relu(omodels.batch_norm(yr, gd, bd, rmr, rvr, True, 0.1, 1e-5))
    want = F.max_pool2d(zr, 3, 2, 1).permute(0, 2, 3, 1)
    assert_close(out, want, 1e-3, 2e-3, "stem forward")
    dout, doutr = f32("st/dout", tuple(out.shape), 1.0)
    (out * dout).sum().backward()
    (want * doutr.double()).sum().backward()
    tol = 30 * EPS16[dtype]
    for name, got, ref in (("dw", wg.grad, wd.grad), ("dgamma", gg.grad, gd.grad), ("dbeta", bg.grad, bd.grad)):
        r = rel_rms(got, ref)
        assert r < tol, (name, r)


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv_weight_pack_batched(dtype):
    """eoe_conv_pack_weight_multi (64x64-tile kernel, > 32 jobs = two launches, ragged tiles, channel padding) against the layout
    definition: w16[o, tap*cpad + c] = w[o, c, tap], w16t = its transpose, w16d[c, (taps-1-tap)*cout + o] = w[o, c, tap]"""
    import eoe_amd
    from eoe_amd import ops
    eoe_amd.set_compute_dtype(dtype)
    try:
        shapes = [(64, 64, 3, None), (128, 64, 1, None), (72, 16, 5, None), (32, 3, 5, 8), (8, 8, 3, None), (200, 136, 3, None)]
        shapes += [(16 + 8 * (i % 4), 8 * (1 + i % 3), 3, None) for i in range(34)]
        ws = [torch.nn.Parameter(torch.randn(co, ci, k, k, device="cuda")) for co, ci, k, _ in shapes]
        ops.refresh_conv_weight_copies([(w, cp) if cp else w for w, (_, _, _, cp) in zip(ws, shapes)])
        for w, (co, ci, k, cp) in zip(ws, shapes):
            cpad, taps = cp or ci, k * k
            w16, w16t, w16d = ops._conv_weight_copies(w, cp)                     # served from the refreshed cache
            kp = w16.shape[1]
            assert kp % 64 == 0 and kp >= taps * cpad
            ref = torch.zeros(co, kp, device="cuda")
            ref.view(co, -1)[:, :taps * cpad].view(co, taps, cpad)[:, :, :ci] = w.detach().reshape(co, ci, taps).permute(0, 2, 1)
            ref = ref.to(dtype)
            assert torch.equal(w16, ref), (co, ci, k, cp)
            assert torch.equal(w16t, ref.t().contiguous())
            refd = w.detach().reshape(co, ci, taps).flip(2).permute(1, 2, 0).reshape(ci, taps * co).to(dtype)
            assert torch.equal(w16d, refd), (co, ci, k, cp)
    finally:
        eoe_amd.set_compute_dtype("fp16")


@pytest.mark.parametrize("dtype", ["bf16"])
def test_wideresnet32_graph_replay_equals_eager(dtype):
    """the HIP-graph replay `bench.py --model wrn --res 32` uses by default (a launch-bound ~400-kernel step): 20 replayed Adam steps
    against the eager run at the benchmark batch.  The WideResNet step holds no atomics, so the trajectories have identical bits."""
    import eoe_amd
    from eoe_amd import parallel
    from eoe_amd.models import WideResNet
    eoe_amd.set_compute_dtype(dtype)
    dev, nb = torch.device("cuda"), 128
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234)
    imgs = torch.randn((2 * nb, 3, 32, 32), generator=gen, device=dev)
    imgs[nb:] += 0.5
    lbls = torch.cat([torch.zeros(nb, dtype=torch.int64), torch.ones(nb, dtype=torch.int64)]).to(dev)

    def run(graph):
        torch.manual_seed(0)
        model = WideResNet(res=32).to(dev).train()
        opt = eoe_amd.FusedAdam(model.parameters(), lr=1e-3, weight_decay=0.0)
        arena = parallel.GradArena(model)
        gs = eoe_amd.GraphedStep(model, lambda f, y: eoe_amd.hsc_loss(f, y, 0, 1.0 / (2 * nb)), eoe_amd.hsc_score, imgs, lbls) if graph else None
        losses = []
        for _ in range(20):
            opt.zero_grad()
            if graph:
                loss, _ = gs(imgs, lbls)
            else:
                loss = eoe_amd.hsc_loss(model(imgs), lbls, 0, 1.0 / (2 * nb))
                loss.backward()
            opt.step()
            losses.append(loss.detach().clone())
        del arena
        return torch.stack(losses).cpu()
    try:
        ref = run(False)
        assert torch.isfinite(ref).all()
        for rep in range(2):
            got = run(True)
            assert torch.equal(got, ref), (rep, (got != ref).nonzero().flatten()[:4].tolist(), (got - ref).abs().max().item())
    finally:
        eoe_amd.set_compute_dtype("fp16")


@pytest.mark.parametrize("cin,planes,stride,H", [(64, 64, 1, 14), (64, 128, 2, 14), (128, 128, 1, 9)])
def test_cbam_junction_fused_equals_the_two_units(cin, planes, stride, H):
    """round 3: ChannelGate + SpatialGate + residual junction as ONE unit (eoe_cbam_junction_*: the channel-gated tensor and the spatial
    gate's input gradient are never written) against the two units of round 2 on a whole BasicBlock: the same fp32 products in the same
    order -- output, input gradient, every parameter gradient and the running buffers agree to the last bits (reductions are summed in
    the same order too; allowance 1e-6 for contraction differences between the kernels)"""
    import copy
    import torch.nn as nn
    import eoe_amd
    from eoe_amd import ops_resnet
    from eoe_amd.models.resnet import BasicBlock
    eoe_amd.set_compute_dtype("fp16")
    eoe_amd.set_parity_mode(True)           # fp32 convolutions around the unit: a last-bit difference is not amplified by a 16-bit rounding
    torch.manual_seed(4)
    ds = stride != 1 or cin != planes
    down = nn.Sequential(nn.Conv2d(cin, planes, 1, stride=stride, bias=False), nn.BatchNorm2d(planes)) if ds else None
    blk0 = BasicBlock(cin, planes, stride, down, use_cbam=True).cuda().train()
    x = torch.randn(6, H, H, cin, device="cuda")
    w = torch.randn(6, (H - 1) // stride + 1, (H - 1) // stride + 1, planes, device="cuda")
    res = {}
    old = ops_resnet.FUSE_CBAM
    try:
        for fused in (False, True):
            ops_resnet.FUSE_CBAM = fused
            blk = copy.deepcopy(blk0)
            xin = x.clone().requires_grad_(True)
            y = blk(xin)
            (y * w).sum().backward()
            res[fused] = (y.detach(), None, xin.grad, {n: p.grad for n, p in blk.named_parameters()},
                          {n: b.clone() for n, b in blk.named_buffers()})
    finally:
        ops_resnet.FUSE_CBAM = old
        eoe_amd.set_parity_mode(False)
    (y0, h0, dx0, g0, b0), (y1, h1, dx1, g1, b1) = res[False], res[True]
    assert torch.equal(y0, y1) or rel_rms(y1, y0.cpu()) < 1e-6
    assert rel_rms(dx1, dx0.cpu()) < 1e-6, rel_rms(dx1, dx0.cpu())
    for n in g0:
        assert rel_rms(g1[n], g0[n].cpu()) < 1e-5, (n, rel_rms(g1[n], g0[n].cpu()))
    for n in b0:
        assert torch.allclose(b1[n].float(), b0[n].float(), rtol=1e-6, atol=1e-7), n
