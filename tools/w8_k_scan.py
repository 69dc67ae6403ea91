#!/usr/bin/env python
"""Per-tile fixed cost against per-k-tile cost of the NT kernels: the same M x N at K = 768 ... 6144 (12 ... 96 k-tiles), operands rotated through
buffer sets, for the eight-wave 256 x 256 kernel (forced) and the launcher's own choice; a straight-line fit time = a + b * k-tiles per kernel.
python tools/w8_k_scan.py"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eoe_amd.ops as ops
from eoe_amd import _lib

M = 12800
for N in (2304, 3072, 768):
    for name, fl in (("eight-wave forced", 1 | 262144), ("launcher's choice", 1)):
        _lib.set_option("nt_flags", fl)
        pts = []
        for K in (768, 1536, 3072, 6144):
            nset = max(2, min(6, int(600e6 // ((M + N) * K * 2))))
            sets = [(torch.randn(M, K, device="cuda").half(), (torch.randn(N, K, device="cuda") * 0.05).half(), torch.empty(M, N, device="cuda", dtype=torch.float16))
                    for _ in range(nset)]
            for a, w, o in sets:
                ops.gemm_nt(a, w, o)
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    for a, w, o in sets:
                        ops.gemm_nt(a, w, o)
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / (3 * nset) * 1e3)
            pts.append((K // 64, best))
            del sets
        x = np.array([p[0] for p in pts], float)
        y = np.array([p[1] for p in pts], float)
        b, a = np.polyfit(x, y, 1)
        tf = [2.0 * M * N * 64 * k / t / 1e6 for k, t in pts]
        print(f"N={N:5d} {name:18s}: " + "  ".join(f"K={int(k * 64):5d} {t:7.1f} us ({f:5.0f} TF)" for (k, t), f in zip(pts, tf)) +
              f"   fit: {a:6.1f} us + {b:5.2f} us per k-tile  (asymptote {2.0 * M * N * 64 / b / 1e6:5.0f} TF)")
_lib.set_option("nt_flags", 1)
