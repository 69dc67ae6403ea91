"""GPU tier, round 5: scheduling changes of the ViT backward sweep that must not change a bit -- the finish reductions collected in one table
and launched once per sweep (eoe_red_table_flush) against one launch per block, and the weight-gradient launches ordered two calls back
(async_wgrad = 2, three buffer sets) against one call back; the C entry point of the table on its own; the box probes of bench.py."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from gpu_util import f32  # noqa: E402
from oracle import models as omodels, trainer as otrainer  # noqa: E402


def _grads(layers, n_half, defer, lag, steps=2):
    import eoe_amd
    from eoe_amd import ops
    from eoe_amd.models import ClipViTB32Custom
    eoe_amd.set_compute_dtype("fp16")
    old = (ops.VIT_DEFER_FINISH, ops.VIT_ASYNC_LAG)
    ops.VIT_DEFER_FINISH, ops.VIT_ASYNC_LAG = defer, lag
    try:
        m = omodels.deterministic_init(ClipViTB32Custom(layers=layers), tag="r5", layers=layers).cuda().train()
        opt = eoe_amd.FusedAdam(m.parameters(), lr=1e-4, weight_decay=1e-3)
        out = []
        for i in range(steps):
            imgs, lbls = otrainer.synthetic_batch(f"r5/b{i}", n_half, n_half, 224)
            opt.zero_grad()
            loss = eoe_amd.hsc_loss(m(imgs.cuda()), lbls.cuda(), 0)
            loss.backward()
            out.append({k: p.grad.detach().clone() for k, p in m.named_parameters()})
            opt.step()
        torch.cuda.synchronize()
        return out
    finally:
        ops.VIT_DEFER_FINISH, ops.VIT_ASYNC_LAG = old


def test_finish_table_and_async_lag_do_not_change_a_bit():
    """a 4-layer tower (three full blocks + the class-token-only last one), 8 + 8 images, two Adam steps: every gradient of both steps bitwise
    equal between {finish per block, lag 1} (round 4's schedule), {one table flush per sweep, lag 1} (the default) and {table, lag 2}"""
    base = _grads(4, 8, False, 1)
    for defer, lag in ((True, 1), (True, 2), (False, 2)):
        got = _grads(4, 8, defer, lag)
        for s, (a, b) in enumerate(zip(base, got)):
            for k in a:
                assert torch.equal(a[k], b[k]), f"defer={defer} lag={lag}: step {s}, gradient {k} differs"


def test_red_table_flush_sums_partial_rows_in_both_layouts():
    """eoe_red_table_flush on its own: 70 jobs (two launches of <= 64), plain [R][N] and blocked [N/64][R][64] partial rows, one to three
    output segments, overwrite and accumulate -- against fp64 sums"""
    from eoe_amd import _lib
    rng = np.random.RandomState(5)
    for overwrite in (1, 0):
        t = _lib.RedTable()
        t.count, t.overwrite = 0, overwrite
        keep, want = [], []
        for j in range(70):
            R = int(rng.choice([1, 7, 64, 160, 200, 512]))
            seg = int(rng.choice([64, 128, 768]))
            nseg = int(rng.choice([1, 2, 3]))
            N = seg * nseg
            blocked = int(j % 2)
            part = torch.randn(R, N, device="cuda")
            ref = part.double().sum(0).cpu()
            if blocked:      # [N/64][R][64]
                stored = part.view(R, N // 64, 64).permute(1, 0, 2).contiguous()
            else:
                stored = part
            outs = [torch.full((seg,), 0.5, device="cuda") for _ in range(nseg)]
            job = t.job[t.count]
            job.part, job.R, job.N, job.seg, job.blocked = stored.data_ptr(), R, N, seg, blocked
            for s in range(3):
                job.out[s] = outs[s].data_ptr() if s < nseg else None
            t.count += 1
            keep += [stored] + outs
            want.append((outs, ref, seg))
        _lib.check(_lib.lib.eoe_red_table_flush(C.byref(t), torch.cuda.current_stream().cuda_stream), "eoe_red_table_flush")
        torch.cuda.synchronize()
        assert t.count == 0
        for outs, ref, seg in want:
            for s, o in enumerate(outs):
                w = ref[s * seg:(s + 1) * seg] + (0.0 if overwrite else 0.5)
                assert torch.allclose(o.double().cpu(), w, rtol=1e-5, atol=1e-4), (overwrite, s)


def test_box_probes_run_and_report_sane_numbers():
    """bench.py's calibration probes through the C ABI: the bare MFMA loop leaves finite sums, the copy kernel copies, and the reported
    rates are inside what an MI355X can do (a probe that silently measured nothing would read 0 or absurdly high)"""
    from eoe_amd import _lib
    s = torch.cuda.current_stream().cuda_stream
    out = torch.full((16 * 256,), float("nan"), device="cuda")
    _lib.check(_lib.lib.eoe_probe_mfma_f16(out.data_ptr(), 10, 16, s), "eoe_probe_mfma_f16")
    assert torch.isfinite(out).all() and out.abs().sum().item() > 0
    src, _ = f32("probe/src", (1 << 18,), 1.0)
    dst = torch.zeros_like(src)
    _lib.check(_lib.lib.eoe_probe_copy(dst.data_ptr(), src.data_ptr(), src.numel() * 4, s), "eoe_probe_copy")
    assert torch.equal(dst, src)
    assert _lib.lib.eoe_probe_copy(dst.data_ptr(), src.data_ptr(), 7, s) == 1          # not a multiple of 16 bytes: an error, not a crash
    import bench
    box = bench.box_probe(torch.device("cuda", 0))
    assert 300 < box["mfma_f16_loop_tf"] < 2600 and 1000 < box["hbm_copy_gbs"] < 8200 and 200 < box["gemm_4096_tf"] < 2600
    assert 200 < box["gemm_cfc_tf"] < 2600


def test_streamk_and_splitk_fall_back_without_a_usable_workspace():
    """eoe_gemm_args.sk_workspace is optional: missing, too small or misaligned, the stream-K form and the split-k hint quietly take the plain
    launches -- same results as with the workspace to within the re-association of fp32 partial sums (split k) / bitwise (data-parallel tiles)"""
    from eoe_amd import _lib, ops
    from gpu_util import t16
    dt = torch.float16
    a, _ = t16("r5/ska", (4096, 768), 1.0, dt)
    w, _ = t16("r5/skw", (2304, 768), 0.05, dt)
    good = ops.nt_sk_workspace(a.device)
    outs = []
    old = _lib.set_option("nt_flags", 1 | 262144 | 1048576)          # eight-wave kernel forced, stream-K asked for
    try:
        for ws, nbytes in ((good.data_ptr(), good.numel()), (None, 0), (good.data_ptr(), 4096), (good.data_ptr() + 4, good.numel() - 4)):
            out = torch.full((4096, 2304), float("nan"), dtype=dt, device="cuda")
            g = _lib.GemmArgs(a.data_ptr(), w.data_ptr(), out.data_ptr(), None, None, None, None, 4096, 2304, 768, 768, 768, 2304, 0,
                              _lib.EOE_F16, 0, 0, 0, 1.0)
            g.sk_workspace, g.sk_workspace_bytes = ws, nbytes
            _lib.check(_lib.lib.eoe_gemm_nt(C.byref(g), torch.cuda.current_stream().cuda_stream), "eoe_gemm_nt")
            torch.cuda.synchronize()
            assert torch.isfinite(out).all()
            outs.append(out)
    finally:
        _lib.set_option("nt_flags", old)
    assert torch.equal(outs[1], outs[2]) and torch.equal(outs[1], outs[3])                 # the three fall-backs: data-parallel tiles
    d = ((outs[0].float() - outs[1].float()).abs() / outs[1].float().abs().clamp_min(1.0)).max().item()
    assert d <= 2.5 * 2.0 ** -11, d                                                        # stream-K: within an ulp of them
    assert int(good[:8192].view(torch.int32).abs().sum()) == 0
    # the split-k hint (small M behind a long K): with and without the workspace
    a2, a2r = t16("r5/spa", (256, 3072), 1.0, dt)
    w2, w2r = t16("r5/spw", (768, 3072), 0.05, dt)
    ref = a2r.double() @ w2r.double().t()
    res = []
    for ws, nbytes in ((good.data_ptr(), good.numel()), (None, 0)):
        out = torch.empty((256, 768), dtype=torch.float32, device="cuda")
        g = _lib.GemmArgs(a2.data_ptr(), w2.data_ptr(), out.data_ptr(), None, None, None, None, 256, 768, 3072, 3072, 3072, 768, 0,
                          _lib.EOE_F16, 0, 1, 0, 1.0)
        g.split_k = 1
        g.sk_workspace, g.sk_workspace_bytes = ws, nbytes
        _lib.check(_lib.lib.eoe_gemm_nt(C.byref(g), torch.cuda.current_stream().cuda_stream), "eoe_gemm_nt")
        torch.cuda.synchronize()
        res.append(out.double().cpu())
    for r in res:
        assert torch.allclose(r, ref, rtol=2e-6, atol=2e-5 * 3072 ** 0.5)
