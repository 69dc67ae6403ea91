"""the 160x256x32 two-workgroup NT kernel (nt_flags bit 12) against the default kernels on the wide-N shapes of the ViT step: same bits?  time
(operands rotated through 6 buffer sets: from HBM as in the step)"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eoe_amd.ops as ops
from eoe_amd import _lib

R = 6
M = 12800


def setflags(v):
    _lib.check(_lib.lib.eoe_set_option(b"nt_flags", v), "eoe_set_option")


def timeit(fn, iters=30):
    for i in range(R):
        fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(iters):
        fn(i % R)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


for dt in (torch.float16, torch.bfloat16):
    for name, n, k, epi in (("qkv fwd", 2304, 768, "none"), ("c_fc fwd", 3072, 768, "gelu"), ("gelu' x dY", 3072, 768, "gelub"), ("ragged M", 3072, 768, "gelu")):
        m = M if name != "ragged M" else 12800 - 37
        A = [torch.randn(m, k, device="cuda").to(dt) for _ in range(R)]
        W = torch.randn(n, k, device="cuda").to(dt) * 0.05
        bias = torch.randn(n, device="cuda")
        O = [torch.empty(m, n, device="cuda", dtype=dt) for _ in range(R)]
        P = [torch.empty(m, n, device="cuda", dtype=dt) for _ in range(R)]
        pre = torch.randn(m, n, device="cuda").to(dt)
        cs = torch.zeros(n, device="cuda")
        if epi == "none":
            fn = lambda i: ops.gemm_nt(A[i], W, O[i], bias=bias)
        elif epi == "gelu":
            fn = lambda i: ops.gemm_nt(A[i], W, O[i], bias=bias, epilogue=ops.EPI_GELU, aux_out=P[i])
        else:
            fn = lambda i: ops.gemm_nt(A[i], W, O[i], epilogue=ops.EPI_GELU_BWD, aux=pre, colsum_out=cs)
        res = {}
        for flags in (0, 4096):
            setflags(flags)
            cs.zero_()
            fn(0)
            torch.cuda.synchronize()
            res[flags] = (O[0].clone(), P[0].clone() if epi == "gelu" else None, cs.clone())
            t = timeit(fn)
            print(f"{str(dt)[6:]:9s} {name:12s} {m}x{n}x{k} flags {flags:5d}: {t:6.1f} us ({2.0 * m * n * k / t / 1e6:5.0f} TF)", flush=True)
        same = torch.equal(res[0][0], res[4096][0]) and (epi != "gelu" or torch.equal(res[0][1], res[4096][1]))
        dcs = (res[0][2] - res[4096][2]).abs().max().item() / (res[0][2].abs().max().item() + 1e-30)
        print(f"     outputs bitwise equal: {same}; column sums rel diff {dcs:.1e}", flush=True)
setflags(0)
