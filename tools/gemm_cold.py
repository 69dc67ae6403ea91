#!/usr/bin/env python
"""does a GEMM variant lose when its operands are not in the Infinity Cache?  Each launch is timed alone (hipEvents around ONE
launch) in three states: hot (same GEMM back to back), B cold (a 1 GB fill evicted everything, then A / aux are re-read), all cold.
usage: python tools/gemm_cold.py 0 128   (nt_flags values)"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eoe_amd.ops as ops
from eoe_amd import _lib

variants = [int(v) for v in sys.argv[1:]] or [0, 128]
dt = torch.float16
M = 12800
shapes = [("qkv fwd", M, 2304, 768, "none"), ("out fwd", M, 768, 768, "res"), ("fc fwd", M, 3072, 768, "gelu"),
          ("proj fwd", M, 768, 3072, "res"), ("dproj", M, 3072, 768, "gelub"), ("dfc", M, 768, 3072, "none"),
          ("dout", M, 768, 768, "none"), ("dqkv", M, 768, 2304, "none")]
flush = torch.empty(1 << 28, device="cuda", dtype=torch.float32)        # 1 GiB: four times the Infinity Cache
sink = torch.zeros(1, device="cuda")
tot = {(v, s): 0.0 for v in variants for s in ("hot", "bcold", "cold")}
for name, m, n, k, epi in shapes:
    a = torch.randn(m, k, device="cuda").to(dt)
    b = (torch.randn(n, k, device="cuda") * 0.05).to(dt)
    bias = torch.randn(n, device="cuda")
    touch = [a]
    if epi == "none":
        out = torch.empty(m, n, device="cuda", dtype=dt); fn = lambda: ops.gemm_nt(a, b, out, bias=bias)
    elif epi == "res":
        out = torch.empty(m, n, device="cuda"); res = torch.randn(m, n, device="cuda"); touch.append(res)
        fn = lambda: ops.gemm_nt(a, b, out, bias=bias, epilogue=ops.EPI_RESIDUAL, aux=res)
    elif epi == "gelu":
        out = torch.empty(m, n, device="cuda", dtype=dt); pre = torch.empty(m, n, device="cuda", dtype=dt)
        fn = lambda: ops.gemm_nt(a, b, out, bias=bias, epilogue=ops.EPI_GELU, aux_out=pre)
    else:
        out = torch.empty(m, n, device="cuda", dtype=dt); pre = torch.randn(m, n, device="cuda").to(dt); touch.append(pre)
        fn = lambda: ops.gemm_nt(a, b, out, epilogue=ops.EPI_GELU_BWD, aux=pre)
    line = f"{name:9s} {m:6d}x{n:5d}x{k:5d} {epi:6s}"
    for v in variants:
        _lib.check(_lib.lib.eoe_set_option(b"nt_flags", v), "opt")
        res_t = {}
        for state in ("hot", "bcold", "cold"):
            ts = []
            for rep in range(10):
                if state == "hot":
                    fn()
                else:
                    flush.fill_(float(rep))
                    if state == "bcold":
                        for t in touch:
                            sink += t.view(-1)[::1].sum(dtype=torch.float32) * 0
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); fn(); e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            res_t[state] = float(np.median(ts))
            tot[(v, state)] += res_t[state]
        line += f" | f{v}: hot {res_t['hot']:6.1f} Bcold {res_t['bcold']:6.1f} cold {res_t['cold']:6.1f} us"
    print(line, flush=True)
for v in variants:
    print(f"layer total f{v}: hot {tot[(v, 'hot')]:.1f}  B cold {tot[(v, 'bcold')]:.1f}  all cold {tot[(v, 'cold')]:.1f} us")
