#!/usr/bin/env python
"""per-shape timing of the fp32 (parity-mode) convolution kernels on the WideResNet-224 / CNN32 shapes at the benchmark batch"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eoe_amd import ops, _lib
from eoe_amd._lib import lib, check
if len(sys.argv) > 1:
    _lib.set_option("parity_flags", int(sys.argv[1]))
N = 256
shapes = [("stem 7x7/2 nchw", 3, 64, 224, 7, 2, 3, True), ("3x3 64->64 @56", 64, 64, 56, 3, 1, 1, False), ("3x3/2 64->128", 64, 128, 56, 3, 2, 1, False),
          ("3x3 128 @28", 128, 128, 28, 3, 1, 1, False), ("1x1/2 64->128", 64, 128, 56, 1, 2, 0, False), ("3x3 256 @14", 256, 256, 14, 3, 1, 1, False),
          ("3x3 512 @7", 512, 512, 7, 3, 1, 1, False), ("cnn32 5x5 3->32 nchw", 3, 32, 32, 5, 1, 2, True), ("cnn32 5x5 32->64", 32, 64, 16, 5, 1, 2, False),
          ("cnn32 5x5 64->128", 64, 128, 8, 5, 1, 2, False), ("fc 2048->512", 2048, 512, 1, 1, 1, 0, False),
          ("3x3/2 128->256 @28", 128, 256, 28, 3, 2, 1, False), ("3x3/2 256->512 @14", 256, 512, 14, 3, 2, 1, False),
          ("1x1/2 128->256 @28", 128, 256, 28, 1, 2, 0, False), ("1x1/2 256->512 @14", 256, 512, 14, 1, 2, 0, False),
          # WideResNet at 32 x 32 (BASELINE config 2): 16 -> 8 -> 4 -> 2 -> 1 maps
          ("w32 stem 7x7/2 nchw", 3, 64, 32, 7, 2, 3, True), ("w32 3x3 64 @8", 64, 64, 8, 3, 1, 1, False), ("w32 3x3/2 64->128 @8", 64, 128, 8, 3, 2, 1, False),
          ("w32 3x3 128 @4", 128, 128, 4, 3, 1, 1, False), ("w32 1x1/2 64->128 @8", 64, 128, 8, 1, 2, 0, False), ("w32 3x3/2 128->256 @4", 128, 256, 4, 3, 2, 1, False),
          ("w32 3x3 256 @2", 256, 256, 2, 3, 1, 1, False), ("w32 3x3/2 256->512 @2", 256, 512, 2, 3, 2, 1, False), ("w32 1x1 512 @1 (tap)", 512, 512, 1, 1, 1, 0, False)]
if len(sys.argv) > 2:
    shapes = [s_ for s_ in shapes if sys.argv[2] in s_[0]]
st = torch.cuda.current_stream().cuda_stream
sk = torch.empty(4 << 20, device="cuda")
p = lambda t: None if t is None else t.data_ptr()
def timeit(fn, it=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it
for name, cin, cout, H, k, s, pad, nchw in shapes:
    Ho = (H + 2 * pad - k) // s + 1
    x = torch.randn((N, cin, H, H) if nchw else (N, H, H, cin), device="cuda")
    w = torch.randn(cout, cin, k, k, device="cuda") * 0.05
    dy = torch.randn(N * Ho * Ho, cout, device="cuda")
    y = torch.empty_like(dy); dx = torch.empty(N, H, H, cin, device="cuda"); dw = torch.empty_like(w)
    mean = torch.tensor([0.1, -0.2, 0.05], device="cuda") if nchw else None
    std = torch.tensor([0.9, 1.1, 1.3], device="cuda") if nchw else None
    geo = ops._geo(N, H, H, cin, k, k, s, pad, Ho, Ho)
    nb = int(lib.eoe_conv_f32_wgrad_workspace(geo, cout)); ws = torch.empty(nb // 4, device="cuda")
    fl = 2.0 * N * Ho * Ho * cout * cin * k * k
    pk = (not nchw) and cout % 4 == 0 and cin % 4 == 0 and len(sys.argv) <= 3
    wf = torch.empty(k * k * cin, cout, device="cuda") if pk else None; wd_ = torch.empty(k * k * cout, cin, device="cuda") if pk else None
    if pk: check(lib.eoe_conv_f32_pack_weights(p(w), p(wf), p(wd_), cout, cin, k, k, st), "pk")
    WKF, WKD = p(wf), p(wd_)
    tf = timeit(lambda: check(lib.eoe_conv_f32_fwd(p(x), int(nchw), p(mean), p(std), p(w), None, p(y), geo, cout, p(sk), sk.numel() * 4, WKF, st), "f"))
    tw = timeit(lambda: check(lib.eoe_conv_f32_wgrad(p(x), int(nchw), p(mean), p(std), p(dy), p(dw), geo, cout, p(ws), nb, st), "w"))
    td = 0.0 if nchw else timeit(lambda: check(lib.eoe_conv_f32_dgrad(p(dy), p(w), p(dx), geo, cout, 0, p(sk), sk.numel() * 4, WKD, st), "d"))
    if nchw:
        x4 = torch.empty((N, H, H, 4), device="cuda"); w4 = torch.zeros(cout, 4, k, k, device="cuda"); w4[:, :3] = w; dw4 = torch.empty_like(w4)
        geo4 = ops._geo(N, H, H, 4, k, k, s, pad, Ho, Ho)
        nb4 = int(lib.eoe_conv_f32_wgrad_workspace(geo4, cout)); ws4 = torch.empty(nb4 // 4, device="cuda")
        tp = timeit(lambda: check(lib.eoe_pack_image_nhwc4(p(x), p(mean), p(std), p(x4), N, H, H, st), "p"))
        tf4 = timeit(lambda: check(lib.eoe_conv_f32_fwd(p(x4), 0, None, None, p(w4), None, p(y), geo4, cout, p(sk), sk.numel() * 4, None, st), "f"))
        tw4 = timeit(lambda: check(lib.eoe_conv_f32_wgrad(p(x4), 0, None, None, p(dy), p(dw4), geo4, cout, p(ws4), nb4, st), "w"))
        print(f"{name + ' (NHWC4)':22s} {fl/1e9:7.1f} GF | fwd {tf4:7.3f} ms {fl/tf4/1e9:6.1f} TF | pack  {tp:7.3f} ms               | wgrad {tw4:7.3f} ms {fl/tw4/1e9:6.1f} TF")
    print(f"{name:22s} {fl/1e9:7.1f} GF | fwd {tf:7.3f} ms {fl/tf/1e9:6.1f} TF | dgrad {td:7.3f} ms {(fl/td/1e9 if td else 0):6.1f} TF | wgrad {tw:7.3f} ms {fl/tw/1e9:6.1f} TF")
