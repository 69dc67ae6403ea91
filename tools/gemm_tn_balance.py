#!/usr/bin/env python
"""what would a perfectly balanced wgrad launch buy?  The ViT block's group (216 tiles of 256x128 on 256 CUs, T = 12800) against a
single GEMM of the same flops with exactly 256 tiles and T = 169 k-tiles (what a stream-K split would give every CU)"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eoe_amd.ops as ops
from eoe_amd import _lib

dt = torch.float16
st = torch.cuda.current_stream().cuda_stream


def group(shapes):
    args = (_lib.GemmArgs * len(shapes))()
    keep, fl = [], 0
    for i, (m, n, t) in enumerate(shapes):
        a = torch.randn(t, m, device="cuda").to(dt)
        b = torch.randn(t, n, device="cuda").to(dt)
        out = torch.empty(m, n, device="cuda", dtype=torch.float32)
        keep += [a, b, out]
        args[i] = _lib.GemmArgs(a.data_ptr(), b.data_ptr(), out.data_ptr(), None, None, None, None, m, n, t, m, n, n, 0,
                                ops.dtype_code(dt), 0, 1, 0, 1.0)
        fl += 2.0 * m * n * t
    return (lambda: _lib.check(_lib.lib.eoe_gemm_tn_grouped(args, len(shapes), st), "g")), fl, keep


cases = {"block group, 216 tiles x 200 k-tiles": [(3072, 768, 12800), (768, 3072, 12800), (2304, 768, 12800), (768, 768, 12800)],
         "balanced, 256 tiles x 169 k-tiles": [(8192, 1024, 10816)],
         "one round, 256 tiles x 200 k-tiles": [(8192, 1024, 12800)]}
fns = {k: group(v) for k, v in cases.items()}
times = {k: [] for k in cases}
for rnd in range(10):
    for k, (fn, fl, _) in fns.items():
        fn(); fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record(); torch.cuda.synchronize()
        times[k].append(e0.elapsed_time(e1) / 5 * 1e3)
for k, (fn, fl, _) in fns.items():
    med = float(np.median(times[k]))
    print(f"{k:40s} {med:7.1f} us  {fl / med / 1e6:6.0f} TF")
