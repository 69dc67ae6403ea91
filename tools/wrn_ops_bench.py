"""per-op timing of the WideResNet conv / BN pieces at the N=256 shapes (torch.cuda.Event around 10 repetitions)
   python tools/wrn_ops_bench.py"""
import sys
import torch
sys.path.insert(0, ".")
import eoe_amd
import eoe_amd.ops as ops
from eoe_amd._lib import lib, check

dev = torch.device("cuda")
N = 256


def timeit(fn, reps=10):
    fn(); fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3     # us


SHAPES = [  # cin, H, cout, k, s, p, count
    (64, 56, 64, 3, 1, 1, 4), (64, 56, 128, 3, 2, 1, 1), (128, 28, 128, 3, 1, 1, 3), (64, 56, 128, 1, 2, 0, 1),
    (128, 28, 256, 3, 2, 1, 1), (256, 14, 256, 3, 1, 1, 3), (128, 28, 256, 1, 2, 0, 1), (256, 14, 512, 3, 2, 1, 1),
    (512, 7, 512, 3, 1, 1, 3), (256, 14, 512, 1, 2, 0, 1)]
dt = torch.float16
tot = {"fwd": 0.0, "wgrad": 0.0, "dgrad": 0.0, "bn_stats": 0.0, "bn_fwd": 0.0, "bn_bwd": 0.0}
print("shape                         fwd us (TF)     wgrad us (TF)    dgrad us (TF)   bn_stats us (GB/s)  bn_fwd   bn_bwd")
for cin, H, cout, k, s, p, cnt in SHAPES:
    Ho = (H + 2 * p - k) // s + 1
    M = N * Ho * Ho
    x16 = torch.randn((N, H, H, cin), device=dev).to(dt)
    w = torch.randn((cout, cin, k, k), device=dev) * 0.05
    w16, w16t, w16d = ops._conv_weight_copies(w)
    y = torch.empty((M, cout), device=dev)
    geo = (N, H, H, cin, k, k, s, p, Ho, Ho)
    fl = 2.0 * M * cout * k * k * cin
    t_f = timeit(lambda: ops.conv_gemm_fwd(x16, w16, y, geo))
    dy16 = torch.randn((M, cout), device=dev).to(dt)
    gT = torch.empty((k * k * cin, cout), device=dev)
    t_w = timeit(lambda: ops.conv_gemm_wgrad(x16, dy16, gT, geo))
    dx = torch.empty((N * H * H, cin), device=dev)
    if s == 1:
        t_d = timeit(lambda: ops.conv_gemm_fwd(dy16.view(N, Ho, Ho, cout), w16d, dx, (N, Ho, Ho, cout, k, k, 1, k - 1 - p, H, H)))
    else:
        kp = w16.shape[1]
        dpat = torch.empty((M, kp), dtype=dt, device=dev)
        dx4 = dx.view(N, H, H, cin)
        def dg():
            ops.gemm_nt(dy16, w16t, dpat)
            check(lib.eoe_col2im(dpat.data_ptr(), dx4.data_ptr(), N, cin, H, H, k, k, s, p, kp, 1, 0, ops._stream()), "col2im")
        t_d = timeit(dg)
    stats = torch.empty(2 * cout, device=dev)
    sums = ops.scratch("bn_sums", (ops.BN_SCRATCH * cout,), torch.float32, dev)
    t_s = timeit(lambda: check(lib.eoe_bn_stats(y.data_ptr(), sums.data_ptr(), stats.data_ptr(), None, None, None, M, cout, 1e-5, 0.1, 1,
                                                ops._stream()), "bn_stats"))
    g = torch.ones(cout, device=dev); b = torch.zeros(cout, device=dev)
    out = torch.empty((N, Ho, Ho, cout), device=dev)
    t_bf = timeit(lambda: check(lib.eoe_bn_act_pool_fwd(y.data_ptr(), stats.data_ptr(), g.data_ptr(), b.data_ptr(), out.data_ptr(), None, N, Ho, Ho,
                                                        cout, 1, 0, 1, 0.0, 1, ops._stream()), "bn_fwd"))
    red = ops.scratch("bn_red", (ops.BN_SCRATCH * cout,), torch.float32, dev)
    dg_, db_ = torch.empty(cout, device=dev), torch.empty(cout, device=dev)
    t_bb = timeit(lambda: check(lib.eoe_bn_act_pool_bwd(y.data_ptr(), stats.data_ptr(), g.data_ptr(), b.data_ptr(), out.data_ptr(),
                                                        red.data_ptr(), dy16.data_ptr(), 0, dg_.data_ptr(), db_.data_ptr(), N, Ho, Ho, cout,
                                                        1, 0, 1, 0, 0.0, 1, ops._stream()), "bn_bwd"))
    by = 4.0 * M * cout
    print(f"{cin:4d}x{H:3d}->{cout:4d} k{k} s{s} x{cnt}   {t_f:7.1f} ({fl / t_f / 1e6:5.0f})  {t_w:7.1f} ({fl / t_w / 1e6:5.0f})  "
          f"{t_d:7.1f} ({fl / t_d / 1e6:5.0f})  {t_s:7.1f} ({by / t_s / 1e3:5.0f})  {t_bf:7.1f} ({2 * by / t_bf / 1e3:5.0f})  "
          f"{t_bb:7.1f} ({2.5 * by / t_bb / 1e3:5.0f})")
    for kname, v in (("fwd", t_f), ("wgrad", t_w), ("dgrad", t_d), ("bn_stats", t_s), ("bn_fwd", t_bf), ("bn_bwd", t_bb)):
        tot[kname] += v * cnt
print("totals over the network (ms, without the stem):", {k: round(v / 1e3, 2) for k, v in tot.items()})
# stem (packed first layer: NHWC4 image with physical padding, gathered inside the GEMMs)
x = torch.randn((N, 3, 224, 224), device=dev)
w = torch.randn((64, 3, 7, 7), device=dev) * 0.05
w16s = ops._stem_weight_copy(w)
M = N * 112 * 112
img = torch.empty((N, 230, 230, 4), dtype=dt, device=dev)
y = torch.empty((M, 64), device=dev)
dy16 = torch.randn((M, 64), device=dev).to(dt)
gT = torch.empty((256, 64), device=dev)
geo = (N, 230, 230, 4, 7, 7, 2, 0, 112, 112)
t_i = timeit(lambda: check(lib.eoe_stem_pack_image(x.data_ptr(), None, None, img.data_ptr(), N, 224, 224, 230, 230, 3, 4, 1, ops._stream()), "pack"))
t_f = timeit(lambda: ops.conv_gemm_fwd(img, w16s, y, geo, mode=2))
t_w = timeit(lambda: ops.conv_gemm_wgrad(img, dy16, gT, geo, mode=2))
stats = torch.empty(128, device=dev); sums = ops.scratch("bn_sums", (ops.BN_SCRATCH * 64,), torch.float32, dev)
t_s = timeit(lambda: check(lib.eoe_bn_stats(y.data_ptr(), sums.data_ptr(), stats.data_ptr(), None, None, None, M, 64, 1e-5, 0.1, 1, ops._stream()), "bn"))
fl = 2.0 * M * 64 * 147
print(f"stem: pack {t_i:.0f} us  fwd {t_f:.0f} us ({fl / t_f / 1e6:.0f} TF)  wgrad {t_w:.0f} us ({fl / t_w / 1e6:.0f} TF)"
      f"  bn_stats {t_s:.0f} us ({4.0 * M * 64 / t_s / 1e3:.0f} GB/s)")
