"""GPU tier 2: every HIP op, called through the C ABI (ctypes), against the CPU oracle / fp64 math on the same
seeded inputs; edge cases: ragged M / N / T tails, sequence shorter than the MFMA tile, accumulate, both dtypes."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from gpu_util import DTYPES, EPS16, t16, f32, assert_close, rel_rms   # noqa: E402
from oracle import objectives, optim as ooptim, models as omodels, fill  # noqa: E402


@pytest.fixture(scope="module")
def ops():
    import eoe_amd.ops as o
    return o


# tile shapes of the NT GEMM (nt_flags of gemm.hip; bit 0 = the shipped fast epilogue): the launcher's own choice (which includes
# the 256x64 two-workgroup kernel for N <= 64), the persistent 256-row kernel (also for N <= 64: bit 6), and the two-workgroup
# kernel with 128- and with 160-row tiles, and the one-wave-per-SIMD kernel (gemm256.hip) with 160x256 and 256x256 tiles -- every
# test below must hold for each
# (round 5) "eight_wave": gemm_w8.hip forced wherever it applies (bit 18), "eight_wave_stream_k": the same in its stream-K form (bit 20), "halves_off":
# the 160x256 kernel's tile order without the column halves (bit 19), "nt128w_old_image": its round-3 LDS swizzle (bit 23)
NT_VARIANTS = {"auto": 1, "persistent256": 1 | 4 | 64, "two_wg_128": 1 | 8 | 16, "two_wg_160": 1 | 8 | 32,
               "one_wave_160x256": 1 | 128, "one_wave_256x256": 1 | 256, "one_wave_cost_model": 1 | 512,
               "eight_wave": 1 | 262144, "eight_wave_stream_k": 1 | 262144 | 1048576, "halves_off": 1 | 524288, "nt128w_old_image": 1 | 8388608}


@pytest.fixture(params=list(NT_VARIANTS))
def nt_variant(request):
    from eoe_amd import _lib
    _lib.check(_lib.lib.eoe_set_option(b"nt_flags", NT_VARIANTS[request.param]), "eoe_set_option")
    yield request.param
    _lib.check(_lib.lib.eoe_set_option(b"nt_flags", NT_VARIANTS["auto"]), "eoe_set_option")


def _qgelu(x):
    return x * torch.sigmoid(1.702 * x)


def _qgelu_grad(x):
    s = torch.sigmoid(1.702 * x)
    return s * (1 + 1.702 * x * (1 - s))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 128), (200, 256, 192), (130, 136, 64), (50, 8, 64),
                                   (12800, 768, 768), (392, 768, 3072), (4, 256, 512), (161, 264, 64), (480, 136, 256), (1000, 64, 192), (300, 48, 64)])
def test_gemm_nt_plain(ops, dtype, M, N, K, nt_variant):
    a, ar = t16(f"nt/a{M}", (M, K), 1.0, dtype)
    b, br = t16(f"nt/b{N}", (N, K), 1.0, dtype)
    bias, biasr = f32("nt/bias", (N,), 1.0)
    ref = ar.double() @ br.double().t() + biasr.double()
    out = torch.empty((M, N), dtype=torch.float32, device="cuda")
    ops.gemm_nt(a, b, out, bias=bias)
    assert_close(out, ref, 2e-6, 2e-5 * math.sqrt(K), f"gemm_nt f32 out {M}x{N}x{K}")
    out16 = torch.empty((M, N), dtype=dtype, device="cuda")
    ops.gemm_nt(a, b, out16, bias=bias)
    assert_close(out16, ref, 2 * EPS16[dtype], 1e-4 * math.sqrt(K), f"gemm_nt 16-bit out {M}x{N}x{K}")
    # accumulate + alpha
    base, baser = f32("nt/base", (M, N), 1.0)
    ops.gemm_nt(a, b, base, accumulate=True, alpha=0.5)
    assert_close(base, baser.double() + 0.5 * (ar.double() @ br.double().t()), 2e-6, 2e-5 * math.sqrt(K), "gemm_nt accumulate")


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_nt_strided_and_epilogues(ops, dtype, nt_variant):
    M, N, K = 200, 256, 128
    a, ar = t16("nte/a", (M, K + 64), 1.0, dtype)          # row stride larger than K
    b, br = t16("nte/b", (N, K), 0.2, dtype)
    bias, biasr = f32("nte/bias", (N,), 0.5)
    res, resr = f32("nte/res", (M, N), 1.0)
    acc = ar[:, :K].double() @ br.double().t() + biasr.double()
    out = torch.empty((M, N), dtype=torch.float32, device="cuda")
    ops.gemm_nt(a[:, :K], b, out, bias=bias, epilogue=ops.EPI_RESIDUAL, aux=res)
    assert_close(out, acc + resr.double(), 2e-6, 1e-4, "residual epilogue")
    pre = torch.empty((M, N), dtype=dtype, device="cuda")
    act = torch.empty((M, N), dtype=dtype, device="cuda")
    ops.gemm_nt(a[:, :K], b, act, bias=bias, epilogue=ops.EPI_GELU, aux_out=pre)
    assert_close(pre, acc, 2 * EPS16[dtype], 1e-4, "gelu epilogue: pre-activation")
    assert_close(act, _qgelu(pre.float().cpu().double()), 2 * EPS16[dtype], 1e-4, "gelu epilogue: activation of the stored pre")
    dz = torch.empty((M, N), dtype=dtype, device="cuda")
    ref = (ar[:, :K].double() @ br.double().t()) * _qgelu_grad(pre.float().cpu().double())
    for atomic in (False, True):           # column sums through partial rows in a workspace (default) / fp32 atomics
        cs = torch.full((N,), 2.0, dtype=torch.float32, device="cuda")
        ops.gemm_nt(a[:, :K], b, dz, epilogue=ops.EPI_GELU_BWD, aux=pre, colsum_out=cs, colsum_atomic=atomic)
        assert_close(dz, ref, 2 * EPS16[dtype], 2e-4, "gelu-backward epilogue")
        assert_close(cs, 2.0 + ref.sum(0), 1e-4, 1e-3, "fused column sums of the epilogue result (M tail masked)")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(300, 64, 128), (1000, 256, 64), (70000, 32, 64)])
def test_gemm_nt_colstats_and_bn_stats_partials(ops, dtype, M, N, K):
    """BatchNorm batch statistics from the GEMM epilogue: per-64-row (sum, sum of squares) partial rows, reduced by
    eoe_bn_stats_partials, equal the statistics eoe_bn_stats computes from the stored output (ragged M tail, > 1024 rows)"""
    from eoe_amd import _lib
    a, ar = t16(f"cs/a{M}", (M, K), 1.0, dtype)
    b, br = t16(f"cs/b{N}", (N, K), 0.3, dtype)
    bias, biasr = f32("cs/bias", (N,), 1.0)
    y = torch.empty((M, N), dtype=torch.float32, device="cuda")
    R = (M + 63) // 64
    part = torch.full((R, 2, N), float("nan"), dtype=torch.float32, device="cuda")
    ops.gemm_nt(a, b, y, bias=bias, colstats_ws=part)
    yr = y.double().cpu()
    pr = part.double().cpu()
    assert torch.isfinite(pr).all()
    for r in (0, R // 2, R - 1):
        blk = yr[r * 64:(r + 1) * 64]
        assert_close(pr[r, 0], blk.sum(0), 1e-5, 1e-4, "partial sums")
        assert_close(pr[r, 1], (blk * blk).sum(0), 1e-5, 1e-3, "partial sums of squares")
    stats = torch.empty(2 * N, dtype=torch.float32, device="cuda")
    stats_ref = torch.empty(2 * N, dtype=torch.float32, device="cuda")
    sums = torch.empty(ops.BN_SCRATCH * N, dtype=torch.float32, device="cuda")
    rm, rv = torch.zeros(N, device="cuda"), torch.ones(N, device="cuda")
    rm2, rv2 = torch.zeros(N, device="cuda"), torch.ones(N, device="cuda")
    nbt, nbt2 = torch.zeros(1, dtype=torch.int64, device="cuda"), torch.zeros(1, dtype=torch.int64, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    _lib.check(_lib.lib.eoe_bn_stats_partials(part.data_ptr(), R, sums.data_ptr(), stats.data_ptr(), rm.data_ptr(), rv.data_ptr(),
                                              nbt.data_ptr(), M, N, 1e-5, 0.1, s), "eoe_bn_stats_partials")
    _lib.check(_lib.lib.eoe_bn_stats(y.data_ptr(), sums.data_ptr(), stats_ref.data_ptr(), rm2.data_ptr(), rv2.data_ptr(),
                                     nbt2.data_ptr(), M, N, 1e-5, 0.1, 1, s), "eoe_bn_stats")
    assert_close(stats[:N], yr.mean(0), 1e-5, 1e-5, "mean")
    assert_close(stats[N:], 1.0 / torch.sqrt(yr.var(0, unbiased=False) + 1e-5), 1e-4, 1e-5, "rstd")
    assert_close(stats, stats_ref.double().cpu(), 1e-4, 1e-5, "fused == separate pass")
    assert_close(rm, rm2.double().cpu(), 1e-5, 1e-6, "running mean")
    assert_close(rv, rv2.double().cpu(), 1e-4, 1e-6, "running var")
    assert int(nbt) == 1


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("T,M,N", [(64, 128, 128), (200, 256, 128), (50, 8, 16), (1000, 136, 264), (12800, 768, 768),
                                   (12544, 768, 3072), (4, 256, 512), (4100, 128, 128)])
def test_gemm_tn(ops, dtype, T, M, N):
    a, ar = t16(f"tn/a{T}", (T, M), 1.0, dtype)
    b, br = t16(f"tn/b{T}", (T, N), 1.0, dtype)
    ref = ar.double().t() @ br.double()
    out = torch.full((M, N), 7.0, dtype=torch.float32, device="cuda")
    ops.gemm_tn(a, b, out)
    assert_close(out, ref, 2e-6, 3e-5 * math.sqrt(T), f"gemm_tn {T}x{M}x{N}")
    ops.gemm_tn(a, b, out, accumulate=True)
    assert_close(out, 2 * ref, 4e-6, 6e-5 * math.sqrt(T), f"gemm_tn accumulate {T}x{M}x{N}")


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_tn_grouped(ops, dtype):
    """the four weight gradients of a (small) block in one launch, ragged T, against fp64"""
    import ctypes as C
    from eoe_amd import _lib
    T = 456
    shapes = [(264, 136), (128, 520), (768, 256), (8, 8)]
    args = (_lib.GemmArgs * 4)()
    keep, outs, refs = [], [], []
    for i, (m, n) in enumerate(shapes):
        a, ar = t16(f"tng/a{i}", (T, m), 1.0, dtype)
        b, br = t16(f"tng/b{i}", (T, n), 1.0, dtype)
        out = torch.full((m, n), 3.0, dtype=torch.float32, device="cuda")
        keep += [a, b]
        outs.append(out)
        refs.append(ar.double().t() @ br.double())
        args[i] = _lib.GemmArgs(a.data_ptr(), b.data_ptr(), out.data_ptr(), None, None, None, None, m, n, T, m, n, n, 0,
                                ops.dtype_code(dtype), 0, 1, 0, 1.0)
    _lib.check(_lib.lib.eoe_gemm_tn_grouped(args, 4, torch.cuda.current_stream().cuda_stream), "grouped")
    for out, ref in zip(outs, refs):
        assert_close(out, ref, 2e-6, 3e-5 * math.sqrt(T), "gemm_tn_grouped")
    for i in range(4):
        args[i].accumulate = 1
    _lib.check(_lib.lib.eoe_gemm_tn_grouped(args, 4, torch.cuda.current_stream().cuda_stream), "grouped acc")
    for out, ref in zip(outs, refs):
        assert_close(out, 2 * ref, 4e-6, 6e-5 * math.sqrt(T), "gemm_tn_grouped accumulate")


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_tn_grouped_stream_k(ops, dtype):
    """the ViT block's wgrad group (216 tiles of 256x128 on 256 CUs) with a workspace: the launch runs as #CUs equal k-ranges (stream-K:
    partial tiles handed between workgroups inside the launch).  Ragged T; against fp64, against the one-tile-per-workgroup launch
    (`tn_flags` bit 1), accumulate, and twice over for bitwise repeatability"""
    from eoe_amd import _lib
    T = 64 * 45 + 24
    shapes = [(3072, 768), (768, 3072), (2304, 768), (768, 768)]
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    ws = torch.empty(cus * 256 * 128 * 4, dtype=torch.uint8, device="cuda")
    args = (_lib.GemmArgs * 4)()
    keep, outs, refs = [], [], []
    for i, (m, n) in enumerate(shapes):
        a, ar = t16(f"tnsk/a{i}", (T, m), 1.0, dtype)
        b, br = t16(f"tnsk/b{i}", (T, n), 1.0, dtype)
        out = torch.full((m, n), 3.0, dtype=torch.float32, device="cuda")
        keep += [a, b]
        outs.append(out)
        refs.append((ar.cuda().double().t() @ br.cuda().double()).cpu())
        args[i] = _lib.GemmArgs(a.data_ptr(), b.data_ptr(), out.data_ptr(), None, None, None, None, m, n, T, m, n, n, 0,
                                ops.dtype_code(dtype), 0, 1, 0, 1.0, ws.data_ptr(), ws.numel())
    st = torch.cuda.current_stream().cuda_stream
    ws.fill_(0xFF)                                   # stale partials (NaN patterns) must never be read
    _lib.check(_lib.lib.eoe_gemm_tn_grouped(args, 4, st), "grouped stream-K")
    first = [o.clone() for o in outs]
    for out, ref in zip(outs, refs):
        assert_close(out, ref, 2e-6, 3e-5 * math.sqrt(T), "gemm_tn_grouped stream-K")
    for rep in range(3):
        for o in outs:
            o.fill_(-1.0)
        _lib.check(_lib.lib.eoe_gemm_tn_grouped(args, 4, st), "grouped stream-K again")
        assert all(torch.equal(o, f) for o, f in zip(outs, first)), "stream-K is not bitwise repeatable"
    try:
        _lib.check(_lib.lib.eoe_set_option(b"tn_flags", 2), "opt")
        _lib.check(_lib.lib.eoe_gemm_tn_grouped(args, 4, st), "grouped plain")
    finally:
        _lib.check(_lib.lib.eoe_set_option(b"tn_flags", 0), "opt")
    for o, f in zip(outs, first):                   # same products, the split tiles add three partial sums instead of one
        assert (o - f).abs().max().item() <= 1e-5 * math.sqrt(T) * 4
    for i in range(4):
        args[i].accumulate = 1
    _lib.check(_lib.lib.eoe_gemm_tn_grouped(args, 4, st), "grouped stream-K acc")
    for out, ref in zip(outs, refs):
        assert_close(out, 2 * ref, 4e-6, 6e-5 * math.sqrt(T), "gemm_tn_grouped stream-K accumulate")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", ["block_s2", "ragged_s2", "two_s4"])
def test_gemm_tn_wide_tiles(ops, dtype, case):
    """the wide-tile wgrad kernel (gemm_tn256.hip: 256x256 tiles, the reduction in aligned slices whose workgroups meet inside the
    launch -- last arriver owns the tile): the ViT block's group (108 tiles x 2 slices), a group with ragged M / N / T tails, and a
    two-problem group that runs 4 slices (partials summed from memory in slice order).  Against fp64, against the 256x128 kernel
    (`tn_flags` bit 2), bitwise repeatable whoever arrives last, accumulate and alpha"""
    from eoe_amd import _lib
    shapes, T = {"block_s2": ([(3072, 768), (768, 3072), (2304, 768), (768, 768)], 64 * 50 + 24),
                 "ragged_s2": ([(3072, 768), (768, 3072), (2304, 776), (776, 520)], 64 * 49 + 8),
                 "two_s4": ([(3072, 768), (2304, 768)], 64 * 96 + 40)}[case]
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    ws = torch.empty(cus * 256 * 256 * 4, dtype=torch.uint8, device="cuda")       # EOE_TN_STREAMK_WORKSPACE_BYTES(#CUs)
    n = len(shapes)
    args = (_lib.GemmArgs * n)()
    keep, outs, refs = [], [], []
    for i, (m, nn) in enumerate(shapes):
        a, ar = t16(f"tnw/{case}/a{i}", (T, m), 1.0, dtype)
        b, br = t16(f"tnw/{case}/b{i}", (T, nn), 1.0, dtype)
        out = torch.full((m, nn), 3.0, dtype=torch.float32, device="cuda")
        keep += [a, b]
        outs.append(out)
        refs.append((ar.cuda().double().t() @ br.cuda().double()).cpu())
        args[i] = _lib.GemmArgs(a.data_ptr(), b.data_ptr(), out.data_ptr(), None, None, None, None, m, nn, T, m, nn, nn, 0,
                                ops.dtype_code(dtype), 0, 1, 0, 1.0, ws.data_ptr(), ws.numel())
    st = torch.cuda.current_stream().cuda_stream
    ws.fill_(0xFF)                                   # stale partials (NaN patterns) must never be read
    before = _lib.get_option("tn256_launches")
    _lib.check(_lib.lib.eoe_gemm_tn_grouped(args, n, st), "grouped wide")
    assert _lib.get_option("tn256_launches") == before + 1, "the group did not take the wide-tile kernel"
    first = [o.clone() for o in outs]
    for out, ref in zip(outs, refs):
        assert_close(out, ref, 2e-6, 3e-5 * math.sqrt(T), f"gemm_tn wide {case}")
    for rep in range(4):
        for o in outs:
            o.fill_(-1.0)
        _lib.check(_lib.lib.eoe_gemm_tn_grouped(args, n, st), "grouped wide again")
        assert all(torch.equal(o, f) for o, f in zip(outs, first)), "the wide-tile wgrad is not bitwise repeatable"
    old = _lib.set_option("tn_flags", 8)             # the one-wave-per-SIMD form of the same kernel: same products, same slice meeting -> same bits
    try:
        for o in outs:
            o.fill_(-1.0)
        _lib.check(_lib.lib.eoe_gemm_tn_grouped(args, n, st), "grouped wide, four waves")
        assert all(torch.equal(o, f) for o, f in zip(outs, first)), "four-wave and eight-wave wide-tile wgrad differ"
    finally:
        _lib.set_option("tn_flags", old)
    old = _lib.set_option("tn_flags", 4)             # the 256x128 kernel on the same problems
    try:
        _lib.check(_lib.lib.eoe_gemm_tn_grouped(args, n, st), "grouped 256x128")
    finally:
        _lib.set_option("tn_flags", old)
    for o, f in zip(outs, first):                    # same products, different partial-sum boundaries
        assert (o - f).abs().max().item() <= 1e-5 * math.sqrt(T) * 4
    for i in range(n):
        args[i].accumulate = 1
        args[i].alpha = 0.5
    for o, f in zip(outs, first):
        o.copy_(f)
    _lib.check(_lib.lib.eoe_gemm_tn_grouped(args, n, st), "grouped wide acc")
    for out, ref in zip(outs, refs):
        assert_close(out, 1.5 * ref, 4e-6, 6e-5 * math.sqrt(T), f"gemm_tn wide accumulate {case}")


W8_SHAPES = [(12800, 2304, 768), (12750, 2304, 768), (12763, 3072, 768), (2049, 3072, 768), (2048, 768, 3072), (6400, 2048, 1024)]


def _nt_kinds(ops, kind, a, w, bias, out, pre):
    if kind == "bias":
        ops.gemm_nt(a, w, out, bias=bias)
    elif kind == "nobias":
        ops.gemm_nt(a, w, out)
    elif kind == "gelu":
        ops.gemm_nt(a, w, out, bias=bias, epilogue=ops.EPI_GELU, aux_out=pre)
    elif kind == "gelu_nopre":
        ops.gemm_nt(a, w, out, bias=bias, epilogue=ops.EPI_GELU)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", W8_SHAPES)
def test_gemm_nt_eight_wave(ops, dtype, M, N, K):
    """the eight-wave 256 x 256 kernel (gemm_w8.hip) at the shapes it serves and at ragged M (rows of A past M staged from an out-of-range
    offset, rows of C past M not stored), every epilogue it has (bias / none / GELU pair / GELU without the saved pre-activation): forced
    (nt_flags bit 18) against an fp64 product on sampled rows -- the first and the last rows among them -- and BITWISE against the
    160 x 128 kernel (same products in the same k order, same epilogue arithmetic); the launcher's own choice (auto) equal to both.
    Stream-K form (bit 20): within an ulp of the 16-bit result (fp32 partial sums re-associated), bitwise repeatable, and the workspace's
    ticket / flag words zero again after every launch"""
    from eoe_amd import _lib
    a, ar = t16(f"w8/a{M}x{K}", (M, K), 1.0, dtype)
    w, wr = t16(f"w8/w{N}x{K}", (N, K), 0.05, dtype)
    bias, biasr = f32(f"w8/bias{N}", (N,), 1.0)
    rows = torch.cat([torch.arange(0, 8), torch.arange(M - 8, M), torch.from_numpy(np.random.RandomState(M + N).randint(0, M, 240))])
    acc = ar[rows].double() @ wr.double().t()
    ws_head = ops.nt_sk_workspace(a.device)[:8192].view(torch.int32)
    for kind in ("bias", "nobias", "gelu", "gelu_nopre"):
        res = {}
        for name, f in (("two_wg_160", 1 | 8 | 32), ("eight_wave", 1 | 262144), ("auto", 1), ("stream_k", 1 | 262144 | 1048576), ("stream_k_again", 1 | 262144 | 1048576),
                        ("stream_k_two_tile", 1 | 262144 | 1048576 | 2097152), ("stream_k_first", 1 | 262144 | 1048576 | 4194304)):
            old = _lib.set_option("nt_flags", f)
            try:
                out = torch.full((M, N), float("nan"), dtype=dtype, device="cuda")
                pre = torch.full((M, N), float("nan"), dtype=dtype, device="cuda")
                _nt_kinds(ops, kind, a, w, bias, out, pre)
                torch.cuda.synchronize()
            finally:
                _lib.set_option("nt_flags", old)
            assert int(ws_head.abs().sum()) == 0, f"{name}: ticket / flag words not zero after the launch"
            res[name] = (out, pre)
        out8, pre8 = res["eight_wave"]
        want = acc + (biasr.double() if kind != "nobias" else 0.0)
        if kind in ("bias", "nobias"):
            assert_close(out8[rows.cuda()], want, 2 * EPS16[dtype], 1e-4 * math.sqrt(K), f"eight-wave {kind} {M}x{N}x{K}")
        elif kind == "gelu":
            assert_close(pre8[rows.cuda()], want, 2 * EPS16[dtype], 1e-4 * math.sqrt(K), f"eight-wave GELU pre-activation {M}x{N}x{K}")
            assert torch.equal(pre8, res["two_wg_160"][1])
            # the activation of the ROUNDED pre-activation
            assert_close(out8[rows.cuda()], _qgelu(pre8[rows.cuda()].float().cpu().double()), 2 * EPS16[dtype], 1e-4, "eight-wave GELU activation")
            gelu_out = out8
        else:
            assert torch.equal(out8, gelu_out), "GELU without the saved pre-activation: a different activation"
        assert torch.equal(out8, res["two_wg_160"][0]), f"eight-wave {kind}: not bitwise the 160x128 kernel's result"
        assert torch.equal(res["auto"][0], out8), f"auto {kind}: differs from the forced kernels"
        # stream-K forms: the same products, partial sums of the split tiles added in segment order
        ulps = 4.5 if kind.startswith("gelu") else 2.5       # (the activation of a pre-activation one ulp away: up to two ulps)
        for name in ("stream_k", "stream_k_two_tile", "stream_k_first"):
            o = res[name][0].float()
            assert torch.isfinite(o).all()
            d = ((o - out8.float()).abs() / out8.float().abs().clamp_min(1.0)).max().item()
            assert d <= ulps * EPS16[dtype], (name, kind, d)
        assert torch.equal(res["stream_k"][0], res["stream_k_again"][0]), "stream-K: not bitwise repeatable"
        if kind == "gelu":
            assert torch.equal(res["stream_k"][1], res["stream_k_again"][1])


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_nt_wide_kernel_tile_order_and_image(ops, dtype):
    """the 160 x 256 x 32 two-workgroup kernel at the shape it serves (M = 12 800, N = 3072, K = 768; GELU pair and GELU' x dY with fused
    column sums): the column-halves tile order (nt_flags bit 19 switches it off) and the LDS image swizzled for the ds_read_b128 lane
    groups (bit 23: the round-3 image) change where and when, not what: bitwise the same results, which equal the 160 x 128 kernel's"""
    from eoe_amd import _lib
    M, N, K = 12800, 3072, 768
    a, ar = t16("wk/a", (M, K), 1.0, dtype)
    w, wr = t16("wk/w", (N, K), 0.05, dtype)
    bias, _ = f32("wk/bias", (N,), 1.0)
    res = {}
    for name, f in (("default", 1), ("halves_off", 1 | 524288), ("old_image", 1 | 8388608), ("two_wg_160", 1 | 8 | 32)):
        old = _lib.set_option("nt_flags", f)
        try:
            act = torch.empty((M, N), dtype=dtype, device="cuda")
            pre = torch.empty((M, N), dtype=dtype, device="cuda")
            ops.gemm_nt(a, w, act, bias=bias, epilogue=ops.EPI_GELU, aux_out=pre)
            dz = torch.empty((M, N), dtype=dtype, device="cuda")
            cs = torch.zeros(N, dtype=torch.float32, device="cuda")
            ops.gemm_nt(a, w, dz, epilogue=ops.EPI_GELU_BWD, aux=pre, colsum_out=cs)
            torch.cuda.synchronize()
        finally:
            _lib.set_option("nt_flags", old)
        res[name] = (act, pre, dz, cs)
    rows = torch.arange(0, M, 97)
    acc = ar[rows].double() @ wr.double().t()
    assert_close(res["default"][1][rows.cuda()], acc + bias.cpu().double(), 2 * EPS16[dtype], 1e-4 * math.sqrt(K), "wide kernel: pre-activation")
    for name in ("halves_off", "old_image", "two_wg_160"):
        for i, what in enumerate(("activation", "pre-activation", "GELU' x dY")):
            assert torch.equal(res[name][i], res["default"][i]), f"{name}: {what} differs"
    for name in ("halves_off", "old_image"):
        assert torch.equal(res[name][3], res["default"][3]), f"{name}: column sums differ"


def test_gemm_rejects_bad_shapes(ops):
    a = torch.zeros((8, 40), dtype=torch.bfloat16, device="cuda")
    b = torch.zeros((8, 40), dtype=torch.bfloat16, device="cuda")
    out = torch.zeros((8, 8), dtype=torch.float32, device="cuda")
    # the C ABI rejects a reduction length that is not a multiple of the 64-deep MFMA k-tile ...
    from eoe_amd import _lib
    g = _lib.GemmArgs(a.data_ptr(), b.data_ptr(), out.data_ptr(), None, None, None, None, 8, 8, 40, 40, 40, 8, 0, _lib.EOE_BF16, 0, 1, 0, 1.0)
    assert _lib.lib.eoe_gemm_nt(C.byref(g), None) == 1 and b"multiple of 64" in _lib.lib.eoe_last_error()
    # ... the Python wrapper zero-pads it (CNN28's 1568- and 32-long reductions)
    a2, a2r = t16("bad/a", (8, 40), 1.0, torch.bfloat16)
    b2, b2r = t16("bad/b", (8, 40), 1.0, torch.bfloat16)
    ops.gemm_nt(a2, b2, out)
    assert_close(out, a2r.double() @ b2r.double().t(), 2e-6, 2e-5 * math.sqrt(40), "ragged K")
    with pytest.raises((RuntimeError, AssertionError)):
        ops.gemm_nt(a.float(), b, out)          # operands must be 16-bit and of one type


@pytest.mark.parametrize("dtype", DTYPES)
def test_cast_transpose_colsum(ops, dtype):
    w, wr = f32("ct/w", (200, 136), 1.0)
    d, dt = ops.cast_transpose(w, dtype)
    assert torch.equal(d.cpu(), wr.to(dtype)) and torch.equal(dt.cpu(), wr.to(dtype).t().contiguous())
    x, xr = t16("cs/x", (1000, 264), 1.0, dtype)
    out = torch.full((264,), 3.0, dtype=torch.float32, device="cuda")
    ops.colsum(x, out)
    assert_close(out, xr.double().sum(0), 1e-5, 1e-3, "colsum")
    ops.colsum(x, out, accumulate=True)
    assert_close(out, 2 * xr.double().sum(0), 1e-5, 2e-3, "colsum accumulate")
    c = ops.cast16(w, dtype)
    assert torch.equal(c.cpu(), wr.to(dtype))
    big, bigr = f32("ct/big", (1001, 264), 1.0)
    o16 = torch.empty((1001, 264), dtype=dtype, device="cuda")
    cs = torch.full((264,), 1.0, dtype=torch.float32, device="cuda")
    ops.cast_colsum(big, o16, cs)
    assert torch.equal(o16.cpu(), bigr.to(dtype))
    assert_close(cs, bigr.double().sum(0), 1e-5, 1e-3, "cast_colsum")
    w2, w2r = f32("ct/w2", (130, 67), 1.0)          # odd sizes take the scalar transpose kernel
    d2, dt2 = ops.cast_transpose(w2, dtype)
    assert torch.equal(d2.cpu(), w2r.to(dtype)) and torch.equal(dt2.cpu(), w2r.to(dtype).t().contiguous())
    v, vr = f32("ct/v", (1027,), 1.0)
    assert torch.equal(ops.cast16(v, dtype).cpu(), vr.to(dtype))
    # batched refresh of many weight copies in one launch (more jobs than one launch's table holds, ragged 64x64 tile tails)
    ops.set_compute_dtype(dtype)
    try:
        shapes = [(768, 768), (132, 200), (64, 4), (4, 64), (260, 68)] + [(8 + 4 * i, 12 + 8 * (i % 5)) for i in range(60)]
        params = [torch.nn.Parameter(f32(f"ct/m{i}", sh, 1.0)[0]) for i, sh in enumerate(shapes)]
        ops.shadow.refresh(params)
        for prm in params:
            hit = ops.shadow.cache[id(prm)]
            want = prm.detach().cpu().to(dtype)
            assert torch.equal(hit[2].cpu(), want) and torch.equal(hit[3].cpu(), want.t().contiguous())
            d, dt = ops.shadow.get(prm, True, True)                 # served from the refreshed cache
            assert d is hit[2] and dt is hit[3]
    finally:
        ops.set_compute_dtype(torch.float16)


@pytest.mark.parametrize("dtype", DTYPES)
def test_patchify(ops, dtype):
    x, xr = f32("pf/x", (3, 3, 64, 64), 1.0)
    mean = torch.tensor([0.1, -0.2, 0.3], device="cuda")
    std = torch.tensor([0.5, 2.0, 1.5], device="cuda")
    for use_norm in (False, True):
        got = ops.patchify(x, 32, mean if use_norm else None, std if use_norm else None, dtype)
        xx = (xr - mean.cpu().view(1, 3, 1, 1)) / std.cpu().view(1, 3, 1, 1) if use_norm else xr
        ref = xx.reshape(3, 3, 2, 32, 2, 32).permute(0, 2, 4, 1, 3, 5).reshape(12, 3072)
        assert_close(got, ref.to(dtype).float(), 2 * EPS16[dtype], 1e-6, f"patchify norm={use_norm}")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,D", [(7, 768), (200, 768), (64, 256), (33, 1024)])
def test_layernorm(ops, dtype, rows, D):
    x, xr = f32("ln/x", (rows, D), 2.0, mean=0.5)
    g, gr = f32("ln/g", (D,), 0.2, mean=1.0)
    b, br = f32("ln/b", (D,), 0.2)
    xr = xr.double().requires_grad_(True)
    gr = gr.double().requires_grad_(True)
    br = br.double().requires_grad_(True)
    yr = omodels.layer_norm(xr, gr, br)
    y32 = torch.empty((rows, D), dtype=torch.float32, device="cuda")
    stats = torch.empty((rows, 2), dtype=torch.float32, device="cuda")
    ops.layernorm_fwd(x, g, b, rows, D, D, y32, stats)
    assert_close(y32, yr, 1e-5, 1e-5, "layernorm fwd f32")
    y16 = torch.empty((rows, D), dtype=dtype, device="cuda")
    ops.layernorm_fwd(x, g, b, rows, D, D, y16, stats)
    assert_close(y16, yr, 2 * EPS16[dtype], 1e-5, "layernorm fwd 16-bit")
    dy, dyr = t16("ln/dy", (rows, D), 1.0, dtype)
    res, resr = f32("ln/res", (rows, D), 1.0)
    (yr * dyr.double()).sum().backward()
    dx = torch.empty((rows, D), dtype=torch.float32, device="cuda")
    dx16 = torch.empty((rows, D), dtype=dtype, device="cuda")
    dg = torch.zeros(D, dtype=torch.float32, device="cuda")
    db = torch.zeros(D, dtype=torch.float32, device="cuda")
    dxs = torch.zeros(D, dtype=torch.float32, device="cuda")
    ops.layernorm_bwd(dy, x, stats, g, rows, D, D, dx, D, dres=res, dx16=dx16, dgamma=dg, dbeta=db, dxsum=dxs)
    assert_close(dxs, (xr.grad + resr.double()).sum(0), 1e-4, 1e-4 * math.sqrt(rows), "layernorm bwd column sums of dx")
    assert_close(dx, xr.grad + resr.double(), 1e-4, 1e-4, "layernorm bwd dx")
    assert_close(dx16, xr.grad + resr.double(), 2 * EPS16[dtype], 1e-4, "layernorm bwd dx16")
    assert_close(dg, gr.grad, 1e-4, 1e-4 * math.sqrt(rows), "layernorm bwd dgamma")
    assert_close(db, br.grad, 1e-4, 1e-4 * math.sqrt(rows), "layernorm bwd dbeta")


def _attn_ref(qkv, n, L, heads):
    D = heads * 64
    q, k, v = qkv.reshape(n, L, 3, heads, 64).permute(2, 0, 3, 1, 4)
    s = (q @ k.transpose(-1, -2)) / 8.0
    return (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3).reshape(n * L, D)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("n,L,heads", [(2, 50, 12), (3, 64, 2), (1, 17, 1), (5, 1, 3)])
def test_attention(ops, dtype, n, L, heads):
    D = heads * 64
    qkv, qkvr = t16(f"att/qkv{L}", (n * L, 3 * D), 1.0, dtype)
    do, dor = t16(f"att/do{L}", (n * L, D), 1.0, dtype)
    qkvr = qkvr.double().requires_grad_(True)
    ref = _attn_ref(qkvr, n, L, heads)
    out = torch.empty((n * L, D), dtype=dtype, device="cuda")
    ops.attn_fwd(qkv, out, n, L, heads)
    # probabilities are rounded to 16 bit before P.V: tolerance of a few 16-bit ulps of the value scale
    assert_close(out, ref, 4 * EPS16[dtype], 6 * EPS16[dtype], f"attention fwd L={L}")
    (ref * dor.double()).sum().backward()
    dqkv = torch.full((n * L, 3 * D), float("nan"), dtype=dtype, device="cuda")
    ops.attn_bwd(qkv, do, dqkv, n, L, heads)
    scale = qkvr.grad.abs().max().item()
    assert_close(dqkv, qkvr.grad, 8 * EPS16[dtype], 8 * EPS16[dtype] * scale, f"attention bwd L={L}")
    assert rel_rms(dqkv, qkvr.grad) < 3 * EPS16[dtype]
    # fused in_proj bias gradient: dbias += column sums of dqkv (from the fp32 accumulators), same dqkv
    db = torch.full((3 * D,), 2.0, dtype=torch.float32, device="cuda")
    dqkv2 = torch.full((n * L, 3 * D), float("nan"), dtype=dtype, device="cuda")
    ops.attn_bwd(qkv, do, dqkv2, n, L, heads, dbias=db)
    assert torch.equal(dqkv2, dqkv)
    assert_close(db, 2.0 + qkvr.grad.sum(0), 1e-3, 8 * EPS16[dtype] * scale * math.sqrt(n * L), f"attention bwd bias sums L={L}")


def test_hsc_bce(ops):
    import eoe_amd
    for n, d in ((16, 256), (5, 100), (300, 256)):
        f, fr = f32(f"hsc/f{n}", (n, d), 0.08)
        y = torch.from_numpy(fill.fill_int(f"hsc/y{n}", (n,), 0, 2))
        fg = f.clone().requires_grad_(True)
        loss = eoe_amd.hsc_loss(fg, y.cuda())
        (loss * 1.5).backward()
        frr = fr.double().requires_grad_(True)
        lref = objectives.hsc_loss(frr, y)
        (lref * 1.5).backward()
        assert abs(loss.item() - lref.item()) <= 2e-6 * max(1, abs(lref.item()))
        assert_close(fg.grad, frr.grad, 1e-5, 1e-9, "hsc grad")
        assert_close(eoe_amd.hsc_score(f), objectives.hsc_score(fr.double()), 1e-5, 1e-7, "hsc score")
        # data-parallel normalisation: sum / global count
        l2 = eoe_amd.hsc_loss(f, y.cuda(), 0, 1.0 / (4 * n))
        assert abs(l2.item() - lref.item() / 4) <= 2e-6
    z = torch.zeros(2, 256, device="cuda")
    assert eoe_amd.hsc_loss(z, torch.zeros(2, dtype=torch.long, device="cuda")).item() == 0.0
    assert abs(eoe_amd.hsc_loss(z, torch.ones(2, dtype=torch.long, device="cuda")).item() - 20.7233) < 1e-3
    for n in (16, 300):
        x, xr = f32(f"bce/x{n}", (n, 1), 3.0)
        y = torch.from_numpy(fill.fill_int(f"bce/y{n}", (n,), 0, 2))
        xg = x.clone().requires_grad_(True)
        loss = eoe_amd.bce_loss(xg, y.cuda())
        loss.backward()
        xrr = xr.double().requires_grad_(True)
        lref = objectives.bce_loss(xrr, y)
        lref.backward()
        assert abs(loss.item() - lref.item()) <= 2e-6 * max(1, abs(lref.item()))
        assert_close(xg.grad, xrr.grad, 1e-5, 1e-9, "bce grad")
        assert_close(eoe_amd.bce_score(x), objectives.bce_score(xr.double()), 1e-5, 1e-7, "bce score")


@pytest.mark.parametrize("M,N,K", [(256, 1, 512), (7, 3, 100), (300, 8, 64)])
def test_linear_small(ops, M, N, K):
    x, xr = f32("ls/x", (M, K), 1.0)
    w, wr = f32("ls/w", (N, K), 0.1)
    b, br = f32("ls/b", (N,), 0.5)
    dy, dyr = f32("ls/dy", (M, N), 1.0)
    xg, wg, bg = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y = ops.linear(xg, wg, bg)
    (y * dy).sum().backward()
    xd, wd, bd = xr.double().requires_grad_(True), wr.double().requires_grad_(True), br.double().requires_grad_(True)
    yr = xd @ wd.t() + bd
    (yr * dyr.double()).sum().backward()
    assert_close(y, yr, 1e-5, 1e-5, "linear_small fwd")
    assert_close(xg.grad, xd.grad, 1e-5, 1e-6, "linear_small dx")
    assert_close(wg.grad, wd.grad, 1e-4, 1e-4, "linear_small dw")
    assert_close(bg.grad, bd.grad, 1e-4, 1e-4, "linear_small db")


@pytest.mark.parametrize("wd", [0.0, 1e-3])
def test_fused_adam(golden, wd):
    import eoe_amd
    g = golden("g8_adam")
    shapes = ((7, 5), (33,), (4, 3, 2))
    ps = [torch.nn.Parameter(torch.from_numpy(fill.fill(f"g8/p{i}", s, std=0.5)).cuda()) for i, s in enumerate(shapes)]
    opt = eoe_amd.FusedAdam(ps, lr=1e-2, weight_decay=wd)
    for t in range(5):
        opt.zero_grad()
        for i, p in enumerate(ps):
            if i == 1 and t in (1, 2):
                p.grad = None
            else:
                p.grad = torch.from_numpy(fill.fill(f"g8/g{i}/t{t}", tuple(p.shape), std=0.1)).cuda()
        opt.step()
    for i, p in enumerate(ps):
        np.testing.assert_allclose(p.detach().cpu().numpy(), g[f"wd{wd}/p{i}"], rtol=3e-6, atol=1e-6)
    # big, unaligned sizes against the oracle
    big = [torch.nn.Parameter(torch.from_numpy(fill.fill(f"ad/p{i}", (n,), std=0.5)).cuda()) for i, n in enumerate((70001, 8192, 3))]
    ref = [p.detach().cpu().clone() for p in big]
    st = ooptim.AdamState(ref)
    opt = eoe_amd.FusedAdam(big, lr=3e-3, weight_decay=wd)
    for t in range(3):
        gs = [torch.from_numpy(fill.fill(f"ad/g{i}/{t}", tuple(p.shape), std=0.1)) for i, p in enumerate(big)]
        for p, gg in zip(big, gs):
            p.grad = gg.cuda()
        opt.step()
        ooptim.adam_step(ref, gs, st, lr=3e-3, weight_decay=wd)
    for p, r in zip(big, ref):
        assert_close(p, r, 3e-6, 1e-6, "fused adam vs oracle")


def test_other_objectives_n4_vs_golden(golden):
    """DSAD / DSVDD / focal heads against the vectors made with the stock torch calls of the reference's
    dsad.py / dsvdd.py / focal.py, plus the data-parallel inv_count convention"""
    import eoe_amd
    g = golden("g9_objectives")
    f = torch.from_numpy(fill.fill("g9/features", (16, 256), std=0.08)).cuda()
    y = torch.from_numpy(fill.fill_int("g9/labels", (16,), 0, 2)).cuda()
    ff = f.clone().requires_grad_(True)
    loss = eoe_amd.dsad_loss(ff, y, 0)
    loss.backward()
    assert abs(loss.item() - float(g["dsad_loss"])) <= 2e-5 * abs(float(g["dsad_loss"]))
    np.testing.assert_allclose(ff.grad.cpu().numpy(), g["dsad_grad"], rtol=2e-4, atol=1e-6)
    c = torch.from_numpy(g["dsvdd_center"]).cuda()
    ff = f.clone().requires_grad_(True)
    loss = eoe_amd.dsvdd_loss(ff, c)
    loss.backward()
    assert abs(loss.item() - float(g["dsvdd_loss"])) <= 2e-5 * abs(float(g["dsvdd_loss"]))
    np.testing.assert_allclose(ff.grad.cpu().numpy(), g["dsvdd_grad"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(eoe_amd.dsvdd_score(f, c).cpu().numpy(), g["dsvdd_scores"], rtol=1e-5)
    x = torch.from_numpy(g["focal_x"]).cuda()
    yy = torch.from_numpy(g["focal_y"]).cuda()
    xx = x.clone().requires_grad_(True)
    loss = eoe_amd.focal_loss(xx, yy)
    loss.backward()
    assert abs(loss.item() - float(g["focal_loss"])) <= 2e-5 * abs(float(g["focal_loss"]))
    np.testing.assert_allclose(xx.grad.cpu().numpy(), g["focal_grad"], rtol=5e-4, atol=1e-7)
    np.testing.assert_allclose(eoe_amd.bce_score(x).cpu().numpy(), g["focal_scores"], rtol=1e-5)
    # inv_count (1 / global batch): a half batch with inv_count 1/16 gives its share of the full-batch loss
    l_half = eoe_amd.dsad_loss(f[:8], y[:8], 0, 1.0 / 16).item() + eoe_amd.dsad_loss(f[8:], y[8:], 0, 1.0 / 16).item()
    assert abs(l_half - float(g["dsad_loss"])) <= 2e-5 * abs(float(g["dsad_loss"]))


@pytest.mark.parametrize("objective", ["dsad", "dsvdd", "focal"])
def test_other_objective_trainers_run(objective):
    """the three extra TRAINER entries run end to end on a small CNN32 task and learn (DSVDD: the loss falls)"""
    from eoe_amd.data import SyntheticAD
    from eoe_amd.models import CNN32
    from eoe_amd.training import TRAINER
    torch.manual_seed(0)
    ds = SyntheticAD(n_train_normal=48, n_oe=16, n_test=32, res=32, shift=1.5, seed=3)
    tr = TRAINER[objective](CNN32(bias=True, clf=(objective == "focal")), dataset=ds, epochs=4, lr=1e-3, wdk=0.0, milestones=[],
                            batch_size=16, classes=["only"])
    _, res = tr.run(run_seeds=1)
    assert np.isfinite(tr.last_losses).all() and tr.last_losses[-1] < tr.last_losses[0]
    if objective != "dsvdd":
        assert res["mean_auc"] > 0.8, res


def test_clip_objective_n2_vs_golden_and_oracle(golden):
    """N2: eoe_clip_fwd / bwd / score against the fixture (reference formulas) and the oracle on larger random inputs"""
    import eoe_amd.ops as o
    g = golden("g10_clip_objective")
    f = torch.from_numpy(fill.fill("g10/features", (24, 512), std=0.4))
    y = torch.from_numpy(fill.fill_int("g10/labels", (24,), 0, 2))
    y[5] = 7
    for mode, T in (("one_vs_rest", 2), ("leave_one_out", 30)):
        t = torch.from_numpy(fill.fill(f"g10/text{T}", (T, 512), std=1.0))
        t = t / t.norm(dim=-1, keepdim=True)
        t = t * 0.25 + 0.75 * t[:1]
        t = t / t.norm(dim=-1, keepdim=True)
        for nominal in (0, 1):
            ff = f.cuda().requires_grad_(True)
            loss = o.clip_loss(ff, y.cuda(), t.cuda(), nominal, mode == "leave_one_out")
            loss.backward()
            assert abs(loss.item() - float(g[f"{mode}/n{nominal}/loss"])) < 2e-5 * max(1.0, abs(float(g[f"{mode}/n{nominal}/loss"])))
            assert_close(ff.grad, torch.from_numpy(g[f"{mode}/n{nominal}/grad"]), 2e-4, 2e-6, f"clip grad {mode} nominal={nominal}")
            assert float(ff.grad[5].abs().max()) == 0.0              # neither label: no loss, no gradient
        assert_close(o.clip_score(f.cuda(), (t * 3.0).cuda()), torch.from_numpy(g[f"{mode}/scores"]), 2e-4, 2e-6, f"clip score {mode}")
    # ragged sizes against the oracle: n not a multiple of 4 rows per workgroup, d = 256, T = 5, upstream gradient scale
    n, d, T = 301, 256, 5
    f = torch.from_numpy(fill.fill("clipx/f", (n, d), std=1.0))
    y = torch.from_numpy(fill.fill_int("clipx/y", (n,), 0, 2))
    t = torch.from_numpy(fill.fill("clipx/t", (T, d), std=1.0))
    t = t / t.norm(dim=-1, keepdim=True) * 0.2 + 0.05
    for loo in (False, True):
        fr = f.clone().requires_grad_(True)
        (objectives.clip_loss(fr, y, t, 0, loo) * 3.0).backward()
        fg = f.cuda().requires_grad_(True)
        lg = o.clip_loss(fg, y.cuda(), t.cuda(), 0, loo)
        (lg * 3.0).backward()
        assert abs(lg.item() - objectives.clip_loss(f, y, t, 0, loo).item()) < 1e-4
        assert_close(fg.grad, fr.grad, 5e-4, 1e-6, f"clip grad vs oracle loo={loo}")


@pytest.mark.parametrize("which", ["adam", "sgd"])
def test_optimisers_unscale_scaled_gradients(which):
    """eoe_amd.set_grad_scale(S): the losses hand out S x the gradient, the optimisers multiply by 1/S before anything else (weight
    decay included) -- parameters and optimiser state after 3 steps are the bits of the unscaled run (S is a power of two)"""
    import eoe_amd
    shapes = ((513, 7), (8192,), (3,))

    def run(scale):
        ps = [torch.nn.Parameter(torch.from_numpy(fill.fill(f"gsc/p{i}", s, std=0.5)).cuda()) for i, s in enumerate(shapes)]
        opt = (eoe_amd.FusedAdam(ps, lr=1e-2, weight_decay=1e-2) if which == "adam"
               else eoe_amd.FusedSGD(ps, lr=1e-2, momentum=0.9, nesterov=True, weight_decay=1e-2))
        eoe_amd.set_grad_scale(scale)
        try:
            for t in range(3):
                for i, p in enumerate(ps):
                    p.grad = torch.from_numpy(fill.fill(f"gsc/g{i}/t{t}", tuple(p.shape), std=0.1)).cuda() * scale
                opt.step()
        finally:
            eoe_amd.set_grad_scale(1.0)
        state = [v.clone() for p in ps for v in opt.state[p].values() if torch.is_tensor(v) and v.is_cuda]
        return [p.detach().clone() for p in ps] + state
    for a, b in zip(run(1.0), run(256.0)):
        assert torch.equal(a, b)
    with pytest.raises(ValueError):
        eoe_amd.set_grad_scale(3.0)


def test_fused_sgd_vs_golden(golden):
    """SGD with Nesterov momentum (ad_trainer.py:380-381) = torch.optim.SGD, incl. a parameter whose gradient appears late"""
    import eoe_amd
    g = golden("g10_clip_objective")
    for wd in (0.0, 1e-3):
        ps = [torch.nn.Parameter(torch.from_numpy(fill.fill(f"g10/p{i}", s, std=0.5)).cuda()) for i, s in enumerate(((7, 5), (33,), (4, 3, 2)))]
        opt = eoe_amd.FusedSGD(ps, lr=1e-2, weight_decay=wd, momentum=0.9, nesterov=True)
        for step in range(5):
            opt.zero_grad()
            for i, p in enumerate(ps):
                p.grad = None if (i == 1 and step in (0, 1)) else torch.from_numpy(fill.fill(f"g10/g{i}/t{step}", tuple(p.shape), std=0.1)).cuda()
            opt.step()
        for i, p in enumerate(ps):
            np.testing.assert_allclose(p.detach().cpu().numpy(), g[f"sgd/wd{wd}/p{i}"], rtol=2e-6, atol=2e-7)


def test_clip_trainer_runs():
    """ADClipTrainer on a small encoder with caller-supplied text features: loss decreases, scores separate the halves"""
    import eoe_amd
    from eoe_amd.training import TRAINER
    from eoe_amd.data import SyntheticAD
    from eoe_amd.models import CNN32
    torch.manual_seed(0)
    model = CNN32(rep_dim=64, bias=True)
    text = torch.nn.functional.normalize(torch.randn(2, 64), dim=-1)
    ds = SyntheticAD(n_train_normal=64, n_oe=64, n_test=64, res=32, shift=1.0, seed=1)
    tr = TRAINER["clip"](model, dataset=ds, epochs=3, lr=1e-2, batch_size=32, text_features=text)
    models, res = tr.run()
    assert np.isfinite(tr.last_losses).all() and tr.last_losses[-1] < tr.last_losses[0]
    # (no AUC bar: 100 x cosine logits saturate the softmax score to exactly 0 / 1 on this toy task, which ties the ranking)
    assert set(res) >= {"mean_auc", "mean_avg_prec", "std_auc", "cls_aucs"} and np.isfinite(res["mean_auc"])
    assert isinstance(tr.center, torch.Tensor) and tuple(tr.center.shape) == (2, 64)
    assert torch.allclose(tr.center.norm(dim=-1).cpu(), torch.ones(2), atol=1e-5)          # prepare_metric normalises (clip.py:62)


def test_auc_ap_on_device_vs_oracle_and_golden(golden):
    """N3: eoe_auc_ap (exact pair counts on the GPU) against the oracle's rank statistic / step-wise AP and the sklearn fixture g7:
    ties, heavy ties, ragged n, a large set, single-class inputs"""
    from eoe_amd import metrics
    from oracle import metrics as ometrics
    g = golden("g7_metrics")
    for i in range(4):
        n, ties = int(g[f"n{i}"]), bool(g[f"ties{i}"])
        s = fill.fill(f"g7/s{i}", (n,), std=1.0)
        if ties:
            s = np.round(s * 4) / 4
        y = fill.fill_int(f"g7/y{i}", (n,), 0, 2)
        y[0], y[1] = 0, 1
        auc, ap = metrics.auc_ap_device(torch.from_numpy(y), torch.from_numpy(s.astype(np.float32)).cuda())
        assert abs(auc - float(g[f"auc{i}"])) < 1e-12 and abs(ap - float(g[f"ap{i}"])) < 1e-12
    rng = np.random.default_rng(3)
    for n, levels in ((1, None), (255, 3), (257, None), (1000, 7), (20011, 50), (20011, None)):
        s = rng.standard_normal(n).astype(np.float32)
        if levels:
            s = np.round(s * levels) / levels
        y = (rng.random(n) < 0.3).astype(np.int64)
        auc, ap = metrics.auc_ap_device(torch.from_numpy(y).cuda(), torch.from_numpy(s).cuda())
        ra, rp = ometrics.roc_auc(y, s), ometrics.average_precision(y, s)
        if np.isnan(ra):
            assert np.isnan(auc)
        else:
            assert abs(auc - ra) < 1e-12, (n, levels, auc, ra)
        if np.isnan(rp):
            assert np.isnan(ap)
        else:
            assert abs(ap - rp) < 1e-10, (n, levels, ap, rp)
    auc, ap = metrics.auc_ap_device(torch.ones(5, dtype=torch.int64), torch.arange(5.0).cuda())
    assert np.isnan(auc) and abs(ap - 1.0) < 1e-15
    auc, ap = metrics.auc_ap_device(torch.zeros(5, dtype=torch.int64), torch.arange(5.0).cuda())
    assert np.isnan(auc) and np.isnan(ap)
