#!/usr/bin/env python
"""where a k-tile of the wide-tile wgrad kernel goes (EOE_GEMM_STAMP=1): wave 0 / wave 1 cycles in the first half (+ DMA wait), at the
barrier, in the second half (incl. LDS-DMA issue)"""
import os, sys
os.environ["EOE_GEMM_STAMP"] = "1"
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eoe_amd.ops as ops
from eoe_amd import _lib
dt = torch.float16
M = 12800
args = (_lib.GemmArgs * 4)()
ws = torch.empty(256 * 256 * 256 * 4, dtype=torch.uint8, device="cuda")
keep, outs = [], []
for i, (m, n, t) in enumerate([(3072, 768, M), (768, 3072, M), (2304, 768, M), (768, 768, M)]):
    a = torch.randn(t, m, device="cuda").to(dt); b = torch.randn(t, n, device="cuda").to(dt)
    out = torch.empty(m, n, device="cuda", dtype=torch.float32)
    keep += [a, b]; outs.append(out)
    args[i] = _lib.GemmArgs(a.data_ptr(), b.data_ptr(), out.data_ptr(), None, None, None, None, m, n, t, m, n, n, 0,
                            ops.dtype_code(dt), 0, 1, 0, 1.0, ws.data_ptr(), ws.numel())
st = torch.cuda.current_stream().cuda_stream
for _ in range(5):
    _lib.check(_lib.lib.eoe_gemm_tn_grouped(args, 4, st), "g")
torch.cuda.synchronize()
n = 216
buf = np.zeros(n * 8, dtype=np.uint64)
_lib.check(_lib.lib.eoe_debug_gemm_stamps(buf.ctypes.data, n * 8), "stamps")
s = buf.reshape(n, 2, 4).astype(np.float64)
nk = 100
for w in (0, 1):
    f, b, sec, tot = (np.median(s[:, w, k]) / nk for k in range(4))
    print(f"wave {w}: per k-tile cycles: first half + DMA wait {f:.0f}, barrier {b:.0f}, second half (+ LDS-DMA issue) {sec:.0f}, total {tot:.0f}  (MFMA issue alone: 2048)")
