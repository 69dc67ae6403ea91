#!/usr/bin/env python
"""An NT kernel variant (NTP_FLAGS, default the forced eight-wave 256 x 256 kernel, nt_flags 262144) against another (NTP_OLD_FLAGS, default that
kernel switched off, 131072): bitwise equality on full and ragged shapes, then interleaved timing with the operands rotated through several
buffer sets (so that they come from HBM, as in the training step).  The same checks run in the suite: tests/test_gpu_ops.py::test_gemm_nt_eight_wave.
Usage: python tools/ntp_check.py [fp16|bf16] [--time-only]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eoe_amd.ops as ops
from eoe_amd import _lib

dt = torch.bfloat16 if "bf16" in sys.argv else torch.float16
NEW = int(os.environ.get("NTP_FLAGS", "262144"))       # the kernel under test: 262144 = the eight-wave 256 x 256 kernel forced
OLD = int(os.environ.get("NTP_OLD_FLAGS", "131072"))   # against: 131072 = that kernel off (the 160 x 128 / 160 x 256 kernels)
assert NEW != OLD, "both flag values select the same kernels: nothing to compare"


def flags(v):
    _lib.check(_lib.lib.eoe_set_option(b"nt_flags", v), "eoe_set_option")


def run(kind, a, w, bias, out, pre):
    if kind == "none":
        ops.gemm_nt(a, w, out, bias=bias)
    elif kind == "gelu":
        ops.gemm_nt(a, w, out, bias=bias, epilogue=ops.EPI_GELU, aux_out=pre)
    elif kind == "gelu_nopre":
        ops.gemm_nt(a, w, out, bias=bias, epilogue=ops.EPI_GELU)
    elif kind == "nobias":
        ops.gemm_nt(a, w, out)


ok = True
if "--time-only" not in sys.argv:
    g = torch.Generator(device="cuda").manual_seed(11)
    for m, n, k in ((12800, 2304, 768), (12800, 3072, 768), (12763, 3072, 768), (2048, 256, 640), (6400, 2048, 1024), (4099, 512, 768), (256 * 50, 768, 3072)):
        a = torch.randn(m, k, device="cuda", generator=g).to(dt)
        w = (torch.randn(n, k, device="cuda", generator=g) * 0.05).to(dt)
        bias = torch.randn(n, device="cuda", generator=g)
        for kind in ("none", "gelu", "gelu_nopre", "nobias"):
            res = {}
            for f in (OLD, NEW):
                flags(f)
                out = torch.full((m, n), float("nan"), device="cuda", dtype=dt)
                pre = torch.full((m, n), float("nan"), device="cuda", dtype=dt)
                run(kind, a, w, bias, out, pre)
                torch.cuda.synchronize()
                res[f] = (out, pre)
            same_o = torch.equal(res[OLD][0], res[NEW][0])
            same_p = kind != "gelu" or torch.equal(res[OLD][1], res[NEW][1])
            nan = torch.isnan(res[NEW][0]).any().item()
            if not (same_o and same_p) or nan:
                ok = False
                d = (res[OLD][0].float() - res[NEW][0].float()).abs()
                bad = (d > 0) | torch.isnan(d)
                rows = bad.any(1).nonzero().flatten()
                cols = bad.any(0).nonzero().flatten()
                print(f"MISMATCH {m}x{n}x{k} {kind}: out equal {same_o} pre equal {same_p} nan {nan}; {int(bad.sum())} elements, "
                      f"rows {rows[:6].tolist()}..{rows[-3:].tolist()} ({len(rows)}), cols {cols[:6].tolist()}..{cols[-3:].tolist()} ({len(cols)}), max {d[~torch.isnan(d)].max().item() if (~torch.isnan(d)).any() else 'nan'}")
            else:
                print(f"ok {m}x{n}x{k} {kind}")
    flags(1)
    print("BITWISE", "PASS" if ok else "FAIL")

# ---- timing: interleaved, operands rotated through NSET buffer sets
NSET = int(os.environ.get("NSET", "6"))
M = 12800
for name, n, k, kind in (("in_proj fwd", 2304, 768, "none"), ("c_fc fwd (GELU pair)", 3072, 768, "gelu")):
    sets = []
    for _ in range(NSET):
        a = torch.randn(M, k, device="cuda").to(dt)
        w = (torch.randn(n, k, device="cuda") * 0.05).to(dt)
        bias = torch.randn(n, device="cuda")
        out = torch.empty(M, n, device="cuda", dtype=dt)
        pre = torch.empty(M, n, device="cuda", dtype=dt)
        sets.append((a, w, bias, out, pre))
    res = {OLD: [], NEW: []}
    for rnd in range(5):
        for f in (OLD, NEW):
            flags(f)
            for s in sets:
                run(kind, *s)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for rep in range(4):
                for s in sets:
                    run(kind, *s)
            e1.record()
            torch.cuda.synchronize()
            res[f].append(e0.elapsed_time(e1) / (4 * NSET) * 1e3)
    flags(1)
    fl = 2.0 * M * n * k
    for f in (OLD, NEW):
        v = sorted(res[f])
        print(f"{name:22s} nt_flags {f:6d}: median {v[len(v) // 2]:7.1f} us  min {v[0]:7.1f}  ({fl / v[len(v) // 2] / 1e6:6.0f} TF)   all {['%.1f' % x for x in res[f]]}")
sys.exit(0 if ok else 1)
