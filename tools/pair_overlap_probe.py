#!/usr/bin/env python
"""Timing-only probe (racy by construction: the consumer does NOT wait for the producer's data) of DESIGN.md section 8.2 item 1: how much does the
pair c_fc forward -> c_proj forward gain if c_proj's workgroups may start under c_fc's last round?  c_fc on the persistent eight-wave kernel
(3 rounds, the third 34 % full) or on the launcher's choice (one round of 160 x 256 tiles), c_proj on a second stream with or without the
dependency.  python tools/pair_overlap_probe.py"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eoe_amd.ops as ops
from eoe_amd import _lib

M, D, H = 12800, 768, 3072
NS = 4
sets = []
for _ in range(NS):
    xn = torch.randn(M, D, device="cuda").half()
    wfc = (torch.randn(H, D, device="cuda") * 0.05).half()
    bfc = torch.randn(H, device="cuda")
    act = torch.empty(M, H, device="cuda", dtype=torch.float16)
    pre = torch.empty(M, H, device="cuda", dtype=torch.float16)
    wpr = (torch.randn(D, H, device="cuda") * 0.02).half()
    bpr = torch.randn(D, device="cuda")
    res = torch.randn(M, D, device="cuda")
    out = torch.empty(M, D, device="cuda")
    sets.append((xn, wfc, bfc, act, pre, wpr, bpr, res, out))
side = torch.cuda.Stream()


def pair(s, fc_flags, overlap):
    xn, wfc, bfc, act, pre, wpr, bpr, res, out = s
    main = torch.cuda.current_stream()
    _lib.set_option("nt_flags", fc_flags)
    if overlap:
        side.wait_stream(main)                  # ordered behind what came BEFORE c_fc only
    ops.gemm_nt(xn, wfc, act, bias=bfc, epilogue=ops.EPI_GELU, aux_out=pre)
    _lib.set_option("nt_flags", 1)
    if overlap:
        with torch.cuda.stream(side):
            ops.gemm_nt(act, wpr, out, bias=bpr, epilogue=ops.EPI_RESIDUAL, aux=res)
        main.wait_stream(side)
    else:
        ops.gemm_nt(act, wpr, out, bias=bpr, epilogue=ops.EPI_RESIDUAL, aux=res)


for name, fl in (("c_fc on the eight-wave kernel", 1 | 262144), ("c_fc on the launcher's choice", 1)):
    for overlap in (False, True):
        for s in sets:
            pair(s, fl, overlap)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                for s in sets:
                    pair(s, fl, overlap)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / (3 * NS) * 1e3)
        print(f"{name:32s} c_proj {'on a second stream, no dependency on c_fc' if overlap else 'behind it on the same stream':46s}: {best:7.1f} us per pair")
_lib.set_option("nt_flags", 1)
