#!/usr/bin/env python
"""interleaved A/B of the NT GEMM tuning switches inside ONE process (build with EOE_AB=1).
usage: python tools/gemm_ab.py 0 1 2 3   (nt_flags values; add 48 / 64 to force the 96- / 128-wide tile)"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eoe_amd.ops as ops
from eoe_amd import _lib

variants = [int(v) for v in sys.argv[1:]] or [0, 1, 2, 3]
dt = torch.float16
M = 12800
shapes = [("qkv fwd", M, 2304, 768, "none"), ("out fwd", M, 768, 768, "res"), ("fc fwd", M, 3072, 768, "gelu"),
          ("proj fwd", M, 768, 3072, "res"), ("dproj", M, 3072, 768, "gelub"), ("dfc", M, 768, 3072, "none"),
          ("dout", M, 768, 768, "none"), ("dqkv", M, 768, 2304, "none"), ("4096^3", 4096, 4096, 4096, "none")]
tot = {v: 0.0 for v in variants}
for name, m, n, k, epi in shapes:
    a = torch.randn(m, k, device="cuda").to(dt)
    b = (torch.randn(n, k, device="cuda") * 0.05).to(dt)
    bias = torch.randn(n, device="cuda")
    if epi == "none":
        out = torch.empty(m, n, device="cuda", dtype=dt); fn = lambda: ops.gemm_nt(a, b, out, bias=bias)
    elif epi == "res":
        out = torch.empty(m, n, device="cuda"); res = torch.randn(m, n, device="cuda")
        fn = lambda: ops.gemm_nt(a, b, out, bias=bias, epilogue=ops.EPI_RESIDUAL, aux=res)
    elif epi == "gelu":
        out = torch.empty(m, n, device="cuda", dtype=dt); pre = torch.empty(m, n, device="cuda", dtype=dt)
        fn = lambda: ops.gemm_nt(a, b, out, bias=bias, epilogue=ops.EPI_GELU, aux_out=pre)
    else:
        out = torch.empty(m, n, device="cuda", dtype=dt); pre = torch.randn(m, n, device="cuda").to(dt)
        fn = lambda: ops.gemm_nt(a, b, out, epilogue=ops.EPI_GELU_BWD, aux=pre)
    times = {v: [] for v in variants}
    for rnd in range(12):
        for v in variants:
            _lib.check(_lib.lib.eoe_set_option(b"nt_flags", v), "opt")
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fn()
            e1.record()
            torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) / 5 * 1e3)
    ref = None
    worst = 0.0
    for v in variants:                       # the variants must compute the same thing
        _lib.check(_lib.lib.eoe_set_option(b"nt_flags", v), "opt")
        out.zero_(); fn(); torch.cuda.synchronize()
        cur = out.float().clone()
        if ref is None:
            ref = cur
        else:
            worst = max(worst, ((cur - ref).abs().max() / ref.abs().max().clamp_min(1e-9)).item())
    line = f"{name:9s} {m:6d}x{n:5d}x{k:5d} {epi:6s} maxdiff {worst:.1e}"
    for v in variants:
        med = float(np.median(times[v]))
        if name != "4096^3":
            tot[v] += med
        line += f" | f{v}: {med:7.1f}us {2.0 * m * n * k / med / 1e6:6.0f}TF"
    print(line)
print("layer total: " + " | ".join(f"f{v}: {tot[v]:.1f}us" for v in variants))
