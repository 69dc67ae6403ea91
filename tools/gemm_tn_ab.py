#!/usr/bin/env python
"""interleaved A/B of the grouped wgrad GEMM tuning switch ("tn_flags") inside one process"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eoe_amd.ops as ops
from eoe_amd import _lib

variants = [int(v) for v in sys.argv[1:]] or [0, 2]
dt = torch.float16
M = 12800
args = (_lib.GemmArgs * 4)()
ws = torch.empty(max(ops.TN_WORKSPACE_BYTES, 256 * 256 * 256 * 4), dtype=torch.uint8, device="cuda")      # tn_flags: 2 = no stream-K, 4 = no wide tiles
keep, fl, outs = [], 0, []
for i, (m, n, t) in enumerate([(3072, 768, M), (768, 3072, M), (2304, 768, M), (768, 768, M)]):
    a = torch.randn(t, m, device="cuda").to(dt)
    b = torch.randn(t, n, device="cuda").to(dt)
    out = torch.empty(m, n, device="cuda", dtype=torch.float32)
    keep += [a, b]; outs.append(out)
    args[i] = _lib.GemmArgs(a.data_ptr(), b.data_ptr(), out.data_ptr(), None, None, None, None, m, n, t, m, n, n, 0,
                            ops.dtype_code(dt), 0, 1, 0, 1.0, ws.data_ptr(), ws.numel())
    fl += 2.0 * m * n * t
st = torch.cuda.current_stream().cuda_stream
fn = lambda: _lib.check(_lib.lib.eoe_gemm_tn_grouped(args, 4, st), "g")
ref = None
times = {v: [] for v in variants}
for rnd in range(10):
    for v in variants:
        _lib.check(_lib.lib.eoe_set_option(b"tn_flags", v), "opt")
        fn(); fn()
        torch.cuda.synchronize()
        if rnd == 0:
            cur = [o.clone() for o in outs]
            if ref is None:
                ref = cur
            else:
                print(f"variant {v} max abs diff vs variant {variants[0]}:", max((c - r).abs().max().item() for c, r in zip(cur, ref)))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record(); torch.cuda.synchronize()
        times[v].append(e0.elapsed_time(e1) / 5 * 1e3)
for v in variants:
    med = float(np.median(times[v]))
    print(f"tn_flags {v}: {med:7.1f} us  {fl / med / 1e6:6.0f} TF")
