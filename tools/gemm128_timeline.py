#!/usr/bin/env python
"""per-CU timeline of the two-workgroup NT GEMM (EOE_GEMM_STAMP=1): when are the two co-resident workgroups of a CU in their
prologue / main loop / epilogue, and how much of the time is at least one of them in its MFMA loop"""
import os, sys
os.environ["EOE_GEMM_STAMP"] = "1"
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eoe_amd.ops as ops
from eoe_amd import _lib

dt = torch.float16
for (name, m, n, k, epi) in [("qkv", 12800, 2304, 768, "none"), ("fc", 12800, 3072, 768, "gelu"), ("dfc", 12800, 768, 3072, "none")]:
    a = torch.randn(m, k, device="cuda").to(dt)
    b = (torch.randn(n, k, device="cuda") * 0.05).to(dt)
    bias = torch.randn(n, device="cuda")
    if epi == "none":
        out = torch.empty(m, n, device="cuda", dtype=dt); fn = lambda: ops.gemm_nt(a, b, out, bias=bias)
    else:
        out = torch.empty(m, n, device="cuda", dtype=dt); pre = torch.empty(m, n, device="cuda", dtype=dt)
        fn = lambda: ops.gemm_nt(a, b, out, bias=bias, epilogue=ops.EPI_GELU, aux_out=pre)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    buf = np.zeros(8192 * 8, dtype=np.uint64)
    _lib.check(_lib.lib.eoe_debug_gemm_stamps(buf.ctypes.data, buf.size), "stamps")
    st = buf.reshape(8192, 8).astype(np.int64)
    st = st[st[:, 0] > 0]
    t0 = st[:, 0].min()
    hw, xcc = st[:, 4], st[:, 5] & 0xf
    cu = (xcc << 16) | (hw & 0xff00)            # XCC | SE/SH/CU bits of HW_ID
    print(f"== {name} {m}x{n}x{k} {epi}: {len(st)} workgroups on {len(set(cu.tolist()))} CUs; span {(st[:, 3].max() - t0)} cycles")
    pro, main, ep = st[:, 1] - st[:, 0], st[:, 2] - st[:, 1], st[:, 3] - st[:, 2]
    print(f"   prologue median {np.median(pro):.0f}  main loop median {np.median(main):.0f} (min {main.min()})  epilogue median {np.median(ep):.0f} (max {ep.max()})")
    print(f"   wave 0 inside the MFMA loop: waiting for the LDS-DMA / fragment reads {np.median(st[:, 6] / np.maximum(main, 1)) * 100:.1f} %, at the barrier {np.median(st[:, 7] / np.maximum(main, 1)) * 100:.1f} %")
    cov = []
    for c in sorted(set(cu.tolist()))[:400]:
        w = st[cu == c]
        w = w[np.argsort(w[:, 0])]
        ev = sorted([(x[1], 1) for x in w] + [(x[2], -1) for x in w])
        busy, depth, last = 0, 0, None
        for t, d in ev:
            if depth > 0:
                busy += t - last
            depth += d
            last = t
        cov.append(busy / (w[:, 3].max() - w[:, 0].min()))
    print(f"   fraction of a CU's span with >= 1 workgroup in its MFMA loop: median {np.median(cov):.2f} (min {min(cov):.2f}, max {max(cov):.2f})")
    c = sorted(set(cu.tolist()))[3]
    w = st[cu == c]
    w = w[np.argsort(w[:, 0])]
    print("   one CU (cycles from kernel start): entry / ready / main end / exit")
    for x in w[:12]:
        print("     ", x[0] - t0, x[1] - t0, x[2] - t0, x[3] - t0)
