#!/usr/bin/env python
"""in-kernel timeline of the persistent NT GEMM (EOE_GEMM_STAMP=1): prologue / main loop / epilogue cycles per tile"""
import os, sys
os.environ["EOE_GEMM_STAMP"] = "1"
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eoe_amd.ops as ops
from eoe_amd import _lib

dt = torch.float16
for (name, m, n, k, epi) in [("qkv", 12800, 2304, 768, "none"), ("out", 12800, 768, 768, "res"), ("fc", 12800, 3072, 768, "gelu"),
                             ("proj", 12800, 768, 3072, "res"), ("4096", 4096, 4096, 4096, "none")]:
    a = torch.randn(m, k, device="cuda").to(dt)
    b = (torch.randn(n, k, device="cuda") * 0.05).to(dt)
    bias = torch.randn(n, device="cuda")
    if epi == "none":
        out = torch.empty(m, n, device="cuda", dtype=dt); fn = lambda: ops.gemm_nt(a, b, out, bias=bias)
    elif epi == "res":
        out = torch.empty(m, n, device="cuda"); res = torch.randn(m, n, device="cuda")
        fn = lambda: ops.gemm_nt(a, b, out, bias=bias, epilogue=ops.EPI_RESIDUAL, aux=res)
    else:
        out = torch.empty(m, n, device="cuda", dtype=dt); pre = torch.empty(m, n, device="cuda", dtype=dt)
        fn = lambda: ops.gemm_nt(a, b, out, bias=bias, epilogue=ops.EPI_GELU, aux_out=pre)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    buf = np.zeros(256 * 16, dtype=np.uint64)
    _lib.check(_lib.lib.eoe_debug_gemm_stamps(buf.ctypes.data, buf.size), "stamps")
    st = buf.reshape(256, 16).astype(np.int64)
    used = st[:, 0] > 0
    st = st[used]
    t0 = st[:, 0].min()
    rel = st[:, :15] - t0
    ntile = min(5, int(((st[:, 2:12] > 0).sum(1).max()) // 2))
    print(f"== {name} {m}x{n}x{k} {epi}: {used.sum()} workgroups, up to {ntile} tiles each (cycles of the 100 MHz-independent shader clock)")
    print(f"   start skew (max entry - min entry): {rel[:, 0].max()}")
    print(f"   prologue (entry -> first operands landed): median {np.median(st[:, 1] - st[:, 0]):.0f}")
    for i in range(ntile):
        ok = st[:, 3 + 2 * i] > 0
        if ok.sum() == 0:
            continue
        prev = st[ok, 1] if i == 0 else st[ok, 1 + 2 * i]
        main = st[ok, 2 + 2 * i] - prev
        ep = st[ok, 3 + 2 * i] - st[ok, 2 + 2 * i]
        print(f"   tile {i}: {ok.sum():3d} wgs  main loop median {np.median(main):7.0f} (min {main.min()}, max {main.max()})  "
              f"epilogue median {np.median(ep):7.0f} (max {ep.max()})")
    tot = (st[:, 2:12].max(1) - st[:, 1])
    print(f"   wave 0: DMA/LDS wait {np.median(st[:, 12] / np.maximum(tot, 1)) * 100:.1f} % of the loop, barrier wait {np.median(st[:, 13] / np.maximum(tot, 1)) * 100:.1f} %;"
          f"   wave 1: {np.median(st[:, 14] / np.maximum(tot, 1)) * 100:.1f} % / {np.median(st[:, 15] / np.maximum(tot, 1)) * 100:.1f} %")
    end = np.where(st[:, 2:12] > 0, st[:, 2:12], 0).max(1) - t0
    print(f"   kernel span: {end.max()} cycles; median workgroup finishes at {np.median(end):.0f}")
