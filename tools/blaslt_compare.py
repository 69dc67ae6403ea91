"""The library GEMM (torch.nn.functional.linear -> hipBLASLt / rocBLAS) against eoe_gemm_nt on the PLAIN shapes of the ViT-B/32 step (no fused
epilogue beyond a bias), operands rotated through 6 buffer sets so that they come from HBM as in the step.  Diagnostic only."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eoe_amd.ops as ops

if os.environ.get("EOE_NT_FLAGS"):          # e.g. 512: every ViT shape on the one-wave 160x256 / 256x256 kernel (gemm256)
    from eoe_amd import _lib
    _lib.check(_lib.lib.eoe_set_option(b"nt_flags", int(os.environ["EOE_NT_FLAGS"])), "eoe_set_option")
dt = torch.float16
M = 12800
shapes = [("qkv fwd (bias)", M, 2304, 768, True), ("fc dgrad", M, 768, 3072, False), ("out dgrad", M, 768, 768, False),
          ("qkv dgrad", M, 768, 2304, False)]
R = 6


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


for name, m, n, k, has_bias in shapes:
    A = [torch.randn(m, k, device="cuda").to(dt) for _ in range(R)]
    W = torch.randn(n, k, device="cuda").to(dt) * 0.05
    bias32 = torch.randn(n, device="cuda") if has_bias else None
    bias16 = bias32.to(dt) if has_bias else None
    O = [torch.empty(m, n, device="cuda", dtype=dt) for _ in range(R)]
    t_eoe = timeit(lambda i: ops.gemm_nt(A[i], W, O[i], bias=bias32))
    t_lib = timeit(lambda i: torch.nn.functional.linear(A[i], W, bias16, ) if True else None)
    t_lib_out = timeit(lambda i: torch.addmm(bias16, A[i], W.t(), out=O[i]) if has_bias else torch.mm(A[i], W.t(), out=O[i]))
    fl = 2.0 * m * n * k
    print(f"{name:16s} {m}x{n}x{k}: eoe_gemm_nt {t_eoe:6.1f} us ({fl / t_eoe / 1e6:5.0f} TF)   library {t_lib:6.1f} us / into out= {t_lib_out:6.1f} us "
          f"({fl / min(t_lib, t_lib_out) / 1e6:5.0f} TF)", flush=True)
