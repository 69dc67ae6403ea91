#!/usr/bin/env python
"""where gemm_ntp.hip's time goes (EOE_GEMM_STAMP=1): wave 0 of every workgroup; operands rotated through 6 buffer sets (cold, as in the step)"""
import os, sys
os.environ["EOE_GEMM_STAMP"] = "1"
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eoe_amd.ops as ops
from eoe_amd import _lib
_lib.set_option("nt_flags", int(os.environ.get("NTP_FLAGS", "16384")))
dt = torch.float16
M = 12800
for name, n, k, epi in (("qkv fwd", 2304, 768, "none"), ("fc fwd", 3072, 768, "gelu")):
    sets = []
    for _ in range(6):
        a = torch.randn(M, k, device="cuda").to(dt); b = (torch.randn(n, k, device="cuda") * 0.05).to(dt); bias = torch.randn(n, device="cuda")
        out = torch.empty(M, n, device="cuda", dtype=dt); pre = torch.empty(M, n, device="cuda", dtype=dt)
        sets.append((a, b, bias, out, pre))
    def fn(s):
        a, b, bias, out, pre = s
        if epi == "none": ops.gemm_nt(a, b, out, bias=bias)
        else: ops.gemm_nt(a, b, out, bias=bias, epilogue=ops.EPI_GELU, aux_out=pre)
    for s in sets: fn(s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(sets[0]); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    nwg = 256
    buf = np.zeros(nwg * 16, dtype=np.uint64)
    _lib.check(_lib.lib.eoe_debug_gemm_stamps(buf.ctypes.data, nwg * 16), "stamps")
    s = buf.reshape(nwg, 16).astype(np.float64)
    s = s[s[:, 7] > 0]
    nk = k // 64
    tiles = s[:, 7]
    med = lambda x: float(np.median(x))
    it = tiles * nk
    print(f"{name:8s} {us:6.1f} us | WGs {len(s)} tiles/WG {tiles.min():.0f}-{tiles.max():.0f} | prologue {med(s[:,0]):.0f} | per k-tile: first {med(s[:,1]/it):.0f} "
          f"wait {med(s[:,2]/it):.0f} barrier {med(s[:,3]/it):.0f} second {med(s[:,4]/it):.0f} (MFMA alone 2 x 512) | open drain {med(s[:,5]):.0f} | "
          f"kernel cycles max {s[:,6].max():.0f} median {med(s[:,6]):.0f} -> {s[:,6].max()/us/1e3:.2f} GHz if the longest WG spans the launch")
