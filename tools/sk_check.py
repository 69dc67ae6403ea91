#!/usr/bin/env python
"""Stream-K form of the eight-wave 256 x 256 NT kernel (gemm_w8.hip, round 5) against the shipped data-parallel kernels:
  1. results: every output within fp32-reassociation distance of the data-parallel result (and of an fp64 product on sampled rows),
     bitwise equal between two runs (the in-order sum), ticket / flag words of the workspace zero again after every launch;
  2. interleaved timing per ViT shape with the operands rotated through several buffer sets (they then come from HBM, as in the step).
nt_flags: 1048576 = stream-K ON (opt-in), 2097152 = two-tile stream-K, 4194304 = stream-K part first, 262144 = force the eight-wave kernel.
Usage: python tools/sk_check.py [fp16|bf16] [--time-only] [--check-only]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eoe_amd.ops as ops
from eoe_amd import _lib

dt = torch.bfloat16 if "bf16" in sys.argv else torch.float16
EPS = 2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11
OFF = 1
SK = 1048576


def flags(v):
    _lib.check(_lib.lib.eoe_set_option(b"nt_flags", v), "eoe_set_option")


def run(kind, a, w, bias, out, pre, res=None, cs=None):
    if kind == "none":
        ops.gemm_nt(a, w, out, bias=bias)
    elif kind == "gelu":
        ops.gemm_nt(a, w, out, bias=bias, epilogue=ops.EPI_GELU, aux_out=pre)
    elif kind == "gelu_nopre":
        ops.gemm_nt(a, w, out, bias=bias, epilogue=ops.EPI_GELU)
    elif kind == "nobias":
        ops.gemm_nt(a, w, out)
    elif kind == "residual":
        ops.gemm_nt(a, w, out, bias=bias, epilogue=ops.EPI_RESIDUAL, aux=res)
    elif kind == "gelu_bwd":
        ops.gemm_nt(a, w, out, epilogue=ops.EPI_GELU_BWD, aux=pre, colsum_out=cs)
    else:
        raise ValueError(kind)


def sync_words_zero():
    ws = ops.nt_sk_workspace(torch.device("cuda", torch.cuda.current_device()))
    return int(ws[:8192].view(torch.int32).abs().sum().item()) == 0


ok = True
KINDS = os.environ.get("SK_KINDS", "none,gelu,nobias").split(",")
VARIANTS = {"sk": 1 | SK | 262144, "sk_two_tile": 1 | SK | 262144 | 2097152, "sk_first": 1 | SK | 262144 | 4194304}
if "--time-only" not in sys.argv:
    g = torch.Generator(device="cuda").manual_seed(11)
    shapes = ((12800, 2304, 768), (12800, 3072, 768), (12763, 3072, 768), (12800, 768, 3072), (12800, 768, 768), (12800, 768, 2304), (2049, 3072, 768),
              (6400, 2048, 1024), (12750, 2304, 768), (2048, 768, 3072), (4099, 512, 768), (12800, 256, 256))
    for m, n, k in shapes:
        a = torch.randn(m, k, device="cuda", generator=g).to(dt)
        w = (torch.randn(n, k, device="cuda", generator=g) * 0.05).to(dt)
        bias = torch.randn(n, device="cuda", generator=g)
        resid = torch.randn(m, n, device="cuda", generator=g)
        rows = torch.randint(0, m, (64,), device="cuda", generator=g)
        rows[0], rows[1] = 0, m - 1
        ref64 = a[rows].double() @ w.double().t()
        for kind in KINDS:
            f32out = kind == "residual"
            odt = torch.float32 if f32out else dt
            outs = {}
            for name, f in [("off", OFF)] + list(VARIANTS.items()) + [("sk_again", VARIANTS["sk"])]:
                flags(f)
                out = torch.full((m, n), float("nan"), device="cuda", dtype=odt)
                pre = torch.full((m, n), float("nan"), device="cuda", dtype=dt)
                cs = torch.zeros(n, device="cuda")
                if kind == "gelu_bwd":
                    pre = (torch.randn(m, n, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))).to(dt)
                run(kind, a, w, bias, out, pre, resid, cs)
                torch.cuda.synchronize()
                outs[name] = (out, pre, cs)
                if not sync_words_zero():
                    ok = False
                    print(f"SYNC WORDS NOT ZERO after {name} {m}x{n}x{k} {kind}")
                    ops.nt_sk_workspace(torch.device("cuda", torch.cuda.current_device()))[:8192].zero_()
            base = outs["off"][0].float()
            scale = base.abs().clamp_min(1.0)
            msg = []
            for name in VARIANTS:
                o = outs[name][0].float()
                nan = torch.isnan(o).any().item()
                d = ((o - base).abs() / scale)
                nd = int((d > 0).sum())
                mx = d.max().item()
                good = (not nan) and mx <= (4e-6 if f32out else (4.5 if kind.startswith('gelu') else 2.5) * EPS)   # (the activation of a pre-activation one ulp away: two ulps)
                if kind == "gelu":
                    dp = ((outs[name][1].float() - outs["off"][1].float()).abs() / outs["off"][1].float().abs().clamp_min(1.0)).max().item()
                    good = good and dp <= 2.5 * EPS
                if kind == "gelu_bwd":
                    dc = ((outs[name][2] - outs["off"][2]).abs() / outs["off"][2].abs().clamp_min(1.0)).max().item()
                    good = good and dc <= 1e-4
                    msg.append(f"colsum dev {dc:.1e}")
                msg.append(f"{name}: {nd} differ, max rel {mx:.2e}{' NAN' if nan else ''}")
                if not good:
                    ok = False
                    bad = (d > (4e-6 if f32out else (4.5 if kind.startswith('gelu') else 2.5) * EPS)) | torch.isnan(d)
                    r_ = bad.any(1).nonzero().flatten()
                    c_ = bad.any(0).nonzero().flatten()
                    print(f"  BAD {name}: {int(bad.sum())} elements; rows {r_[:5].tolist()}..{r_[-3:].tolist()} ({len(r_)}); cols {c_[:5].tolist()}..{c_[-3:].tolist()} ({len(c_)})")
            same = torch.equal(outs["sk"][0], outs["sk_again"][0])
            if not same:
                ok = False
            # fp64 product on sampled rows (plain kinds)
            dev64 = ""
            if kind in ("none", "nobias"):
                want = ref64 + (bias.double() if kind == "none" else 0.0)
                got = outs["sk"][0][rows].double()
                e = ((got - want).abs() / want.abs().clamp_min(1.0)).max().item()
                dev64 = f"; vs fp64 {e:.2e}"
                if e > 2 * EPS:
                    ok = False
            print(f"{m}x{n}x{k} {kind:10s} run-to-run equal {same}{dev64}; " + "; ".join(msg))
    flags(1)
    print("RESULTS", "PASS" if ok else "FAIL")
    if not ok:
        sys.exit(1)
if "--check-only" in sys.argv:
    sys.exit(0)

# ---- timing: interleaved, operands rotated through NSET buffer sets
NSET = int(os.environ.get("NSET", "6"))
M = 12800
TIMED = (("in_proj fwd", 2304, 768, "none"), ("c_fc fwd (GELU pair)", 3072, 768, "gelu"), ("dgrad c_fc (768x3072)", 768, 3072, "nobias"),
         ("dgrad in_proj (768x2304)", 768, 2304, "nobias"), ("dgrad out_proj (768x768)", 768, 768, "nobias"))
if "residual" in KINDS:
    TIMED += (("c_proj fwd (+res)", 768, 3072, "residual"), ("out_proj fwd (+res)", 768, 768, "residual"))
if "gelu_bwd" in KINDS:
    TIMED += (("GELU' x dY + colsum", 3072, 768, "gelu_bwd"),)
TV = {"default (no stream-K)": OFF, "stream-K": 1 | SK, "stream-K two-tile": 1 | SK | 2097152, "stream-K first": 1 | SK | 4194304, "w8 data-parallel": 1 | 262144,
      "old nt128w LDS image": 1 | 8388608}
for name, n, k, kind in TIMED:
    sets = []
    for _ in range(NSET):
        a = torch.randn(M, k, device="cuda").to(dt)
        w = (torch.randn(n, k, device="cuda") * 0.05).to(dt)
        bias = torch.randn(n, device="cuda")
        out = torch.empty(M, n, device="cuda", dtype=torch.float32 if kind == "residual" else dt)
        pre = torch.randn(M, n, device="cuda").to(dt)
        res_ = torch.randn(M, n, device="cuda") if kind == "residual" else None
        cs = torch.zeros(n, device="cuda") if kind == "gelu_bwd" else None
        sets.append((a, w, bias, out, pre, res_, cs))
    res = {v: [] for v in TV}
    for rnd in range(5):
        for v, f in TV.items():
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
            res[v].append(e0.elapsed_time(e1) / (4 * NSET) * 1e3)
    flags(1)
    fl = 2.0 * M * n * k
    for v in TV:
        x = sorted(res[v])
        print(f"{name:26s} {v:22s}: median {x[len(x) // 2]:7.1f} us  min {x[0]:7.1f}  ({fl / x[len(x) // 2] / 1e6:6.0f} TF)")
sys.exit(0)
