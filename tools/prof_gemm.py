#!/usr/bin/env python
"""a few launches of each GEMM shape of a ViT-B/32 block (for rocprofv3 --pmc / --kernel-trace runs)"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eoe_amd.ops as ops
from eoe_amd import _lib

dt = torch.float16
M = 12800
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for (m, n, k) in [(M, 2304, 768), (M, 768, 768), (M, 3072, 768), (M, 768, 3072), (4096, 4096, 4096)]:
    a = torch.randn(m, k, device="cuda").to(dt)
    b = (torch.randn(n, k, device="cuda") * 0.05).to(dt)
    out = torch.empty(m, n, device="cuda", dtype=dt)
    for _ in range(reps):
        ops.gemm_nt(a, b, out)
args = (_lib.GemmArgs * 4)()
keep = []
for i, (m, n, t) in enumerate([(3072, 768, M), (768, 3072, M), (2304, 768, M), (768, 768, M)]):
    a = torch.randn(t, m, device="cuda").to(dt)
    b = torch.randn(t, n, device="cuda").to(dt)
    out = torch.empty(m, n, device="cuda", dtype=torch.float32)
    keep += [a, b, out]
    args[i] = _lib.GemmArgs(a.data_ptr(), b.data_ptr(), out.data_ptr(), None, None, None, None, m, n, t, m, n, n, 0,
                            ops.dtype_code(dt), 0, 1, 0, 1.0)
for _ in range(reps):
    _lib.check(_lib.lib.eoe_gemm_tn_grouped(args, 4, torch.cuda.current_stream().cuda_stream), "g")
torch.cuda.synchronize()
print("done")
