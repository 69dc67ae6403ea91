#!/usr/bin/env python
"""where the one-wave NT kernel's time goes (EOE_GEMM_STAMP=1, nt_flags 129 = 160x256 tiles): wave 0 of every workgroup"""
import os, sys
os.environ["EOE_GEMM_STAMP"] = "1"
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eoe_amd.ops as ops
from eoe_amd import _lib
_lib.set_option("nt_flags", int(sys.argv[1]) if len(sys.argv) > 1 else 129)
dt = torch.float16
M = 12800
shapes = [("qkv fwd", M, 2304, 768, "none"), ("fc fwd", M, 3072, 768, "gelu"), ("dproj", M, 3072, 768, "gelub"), ("dfc", M, 768, 3072, "none"),
          ("out fwd", M, 768, 768, "res")]
for name, m, n, k, epi in shapes:
    a = torch.randn(m, k, device="cuda").to(dt); b = (torch.randn(n, k, device="cuda") * 0.05).to(dt); bias = torch.randn(n, device="cuda")
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
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    nwg = 256
    buf = np.zeros(nwg * 16, dtype=np.uint64)
    _lib.check(_lib.lib.eoe_debug_gemm_stamps(buf.ctypes.data, nwg * 16), "stamps")
    s = buf.reshape(nwg, 16).astype(np.float64)
    s = s[s[:, 6] > 0]
    nk = k // 64
    tiles = s[:, 6]
    med = lambda x: float(np.median(x))
    print(f"{name:8s} {us:6.1f} us | WGs {len(s)} tiles/WG {med(tiles):.0f}-{tiles.max():.0f} | prologue {med(s[:,0]):.0f} | per k-tile: first {med(s[:,1]/(tiles*nk)):.0f} "
          f"barrier {med(s[:,2]/(tiles*nk)):.0f} second {med(s[:,3]/(tiles*nk)):.0f} (MFMA alone {2*640}) | epilogue/tile {med(s[:,4]/tiles):.0f} | "
          f"kernel cycles max {s[:,5].max():.0f} median {med(s[:,5]):.0f} -> {s[:,5].max()/us/1e3:.2f} GHz if the longest WG spans the launch")
