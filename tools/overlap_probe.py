#!/usr/bin/env python
"""can the grouped wgrad GEMM of a ViT block run UNDER the HBM-bound kernels of the next block's backward (LayerNorm
backward x2, attention backward, column sums) when it is launched on a second stream?  sequential vs two-stream time.
Also: wgrad GEMM under a dgrad NT GEMM (both MFMA-bound; only tail effects can overlap)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eoe_amd.ops as ops
from eoe_amd import _lib

dt = torch.float16
M, D, n, L, H = 12800, 768, 256, 50, 12
dev = "cuda"
args = (_lib.GemmArgs * 4)()
keep = []
for i, (m, nn, t) in enumerate([(3072, 768, M), (768, 3072, M), (2304, 768, M), (768, 768, M)]):
    a = torch.randn(t, m, device=dev).to(dt); b = torch.randn(t, nn, device=dev).to(dt)
    out = torch.empty(m, nn, device=dev)
    keep += [a, b, out]
    args[i] = _lib.GemmArgs(a.data_ptr(), b.data_ptr(), out.data_ptr(), None, None, None, None, m, nn, t, m, nn, nn, 0,
                            ops.dtype_code(dt), 0, 1, 0, 1.0)


def tn(stream):
    _lib.check(_lib.lib.eoe_gemm_tn_grouped(args, 4, stream.cuda_stream), "tn")


x = torch.randn(M, D, device=dev); dy = torch.randn(M, D, device=dev)
stats = torch.stack([x.mean(1), 1.0 / x.std(1)], 1).contiguous()
g = torch.ones(D, device=dev)
dx = torch.empty(M, D, device=dev); dx16 = torch.empty(M, D, device=dev, dtype=dt)
dg = torch.zeros(D, device=dev); db = torch.zeros(D, device=dev)
qkv = torch.randn(M, 3 * D, device=dev).to(dt); datt = torch.randn(M, D, device=dev).to(dt); dqkv = torch.empty_like(qkv)
cs = torch.zeros(3 * D, device=dev)
a16 = torch.randn(M, D, device=dev).to(dt); w16 = torch.randn(3072, D, device=dev).to(dt); c16 = torch.empty(M, 3072, device=dev, dtype=dt)


def membound():
    ops.layernorm_bwd(dy, x, stats, g, M, D, D, dx, D, dx16=dx16, dgamma=dg, dbeta=db)
    ops.attn_bwd(qkv, datt, dqkv, n, L, H)
    ops.colsum(dqkv, cs)
    ops.layernorm_bwd(dy, x, stats, g, M, D, D, dx, D, dx16=dx16, dgamma=dg, dbeta=db)


def nt():
    ops.gemm_nt(a16, w16, c16)


main, side = torch.cuda.current_stream(), torch.cuda.Stream()


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def both(work):
    def f():
        side.wait_stream(main)
        tn(side)
        work()
        main.wait_stream(side)
    return f


for name, work in (("HBM-bound chain (LN bwd x2, attn bwd, colsum)", membound), ("dgrad-like NT GEMM 12800x3072x768", nt)):
    t_w = timed(work); t_t = timed(lambda: tn(main)); t_seq = timed(lambda: (tn(main), work())); t_par = timed(both(work))
    print(f"{name}: alone {t_w:.0f} us, wgrad alone {t_t:.0f} us, sequential {t_seq:.0f} us, two streams {t_par:.0f} us "
          f"(hidden {t_seq - t_par:.0f} us = {100 * (t_seq - t_par) / min(t_w, t_t):.0f} % of the shorter)")

# ---- second question: the optimiser (pure HBM streaming, no LDS, few registers) under the MFMA GEMMs
import eoe_amd
w = [torch.randn(3072, 768, device=dev, requires_grad=True) for _ in range(28)]       # ~66 M parameters
for p in w:
    p.grad = torch.randn_like(p)
opt = eoe_amd.FusedAdam(w, lr=1e-4)
opt.step()
a2 = torch.randn(M, 3072, device=dev).to(dt); w2 = torch.randn(768, 3072, device=dev).to(dt); c2 = torch.empty(M, 768, device=dev, dtype=dt)


def gemms():
    for _ in range(3):
        ops.gemm_nt(a16, w16, c16)          # 128x128 two-workgroup kernel (plain epilogue)
        ops.gemm_nt(a2, w2, c2)
    tn(main)


def adam_side():
    with torch.cuda.stream(side):
        opt.step()


def both2():
    side.wait_stream(main)
    adam_side()
    gemms()
    main.wait_stream(side)


t_g = timed(gemms); t_a = timed(lambda: opt.step()); t_seq = timed(lambda: (opt.step(), gemms())); t_par = timed(both2)
print(f"Adam over 66 M parameters under 6 NT GEMMs + the grouped wgrad: GEMMs alone {t_g:.0f} us, Adam alone {t_a:.0f} us, sequential {t_seq:.0f} us, "
      f"two streams {t_par:.0f} us (hidden {t_seq - t_par:.0f} us = {100 * (t_seq - t_par) / min(t_g, t_a):.0f} % of the shorter)")

# ---- third question (round 2): one block's Adam (7 M parameters) under ONE stream-K wgrad launch (all 256 CUs hold a 448-register,
# 144-KB workgroup: 64 registers per SIMD and 16 KB of LDS are left for a co-resident streaming kernel)
ws = torch.empty(ops.TN_WORKSPACE_BYTES, dtype=torch.uint8, device=dev)
args[0].workspace, args[0].workspace_bytes = ws.data_ptr(), ws.numel()
wb = [torch.randn(2304, 768, device=dev, requires_grad=True), torch.randn(768, 768, device=dev, requires_grad=True),
      torch.randn(3072, 768, device=dev, requires_grad=True), torch.randn(768, 3072, device=dev, requires_grad=True)]
for p in wb:
    p.grad = torch.randn_like(p)
optb = eoe_amd.FusedAdam(wb, lr=1e-4)
optb.step()


def both3():
    side.wait_stream(main)
    tn(main)
    with torch.cuda.stream(side):
        optb.step()
    main.wait_stream(side)


for flags in (0, 2):
    _lib.check(_lib.lib.eoe_set_option(b"tn_flags", flags), "opt")
    t_t = timed(lambda: tn(main)); t_a = timed(lambda: optb.step()); t_seq = timed(lambda: (tn(main), optb.step())); t_par = timed(both3)
    print(f"tn_flags {flags} ({'stream-K' if flags == 0 else '216 tiles'}): wgrad alone {t_t:.0f} us, one block's Adam alone {t_a:.0f} us, sequential {t_seq:.0f} us, "
          f"two streams {t_par:.0f} us (hidden {t_seq - t_par:.0f} us)")
_lib.check(_lib.lib.eoe_set_option(b"tn_flags", 0), "opt")
