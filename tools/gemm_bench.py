#!/usr/bin/env python
"""per-shape timing of the GEMM kernels on the shapes of one ViT-B/32 training step (diagnostic tool)"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eoe_amd.ops as ops

if os.environ.get("EOE_NT_FLAGS"):
    from eoe_amd import _lib
    _lib.check(_lib.lib.eoe_set_option(b"nt_flags", int(os.environ["EOE_NT_FLAGS"])), "eoe_set_option")
dt = torch.float16 if (len(sys.argv) < 2 or sys.argv[1] == "fp16") else torch.bfloat16
M = 12800
shapes_nt = [("qkv fwd", M, 2304, 768, "none"), ("out fwd", M, 768, 768, "res"), ("fc fwd", M, 3072, 768, "gelu"),
             ("proj fwd", M, 768, 3072, "res"), ("dproj", M, 3072, 768, "gelub"), ("dfc", M, 768, 3072, "none"),
             ("dout", M, 768, 768, "none"), ("dqkv", M, 768, 2304, "none"), ("patch", 12544, 768, 3072, "f32"),
             ("4096^3", 4096, 4096, 4096, "none")]
shapes_tn = [("w_in", 2304, 768, M), ("w_out", 768, 768, M), ("w_fc", 3072, 768, M), ("w_proj", 768, 3072, M),
             ("w_conv", 768, 3072, 12544), ("4096^3", 4096, 4096, 4096)]


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3   # us


print(f"dtype {dt}  EOE_GEMM_DEBUG={os.environ.get('EOE_GEMM_DEBUG', '0')}")
tot_us = tot_fl = 0
for name, m, n, k, epi in shapes_nt:
    a = torch.randn(m, k, device="cuda").to(dt)
    b = torch.randn(n, k, device="cuda").to(dt) * 0.05
    bias = torch.randn(n, device="cuda")
    if epi == "none":
        out = torch.empty(m, n, device="cuda", dtype=dt)
        fn = lambda: ops.gemm_nt(a, b, out, bias=bias)
    elif epi == "f32":
        out = torch.empty(m, n, device="cuda", dtype=torch.float32)
        fn = lambda: ops.gemm_nt(a, b, out)
    elif epi == "res":
        out = torch.empty(m, n, device="cuda", dtype=torch.float32)
        res = torch.randn(m, n, device="cuda")
        fn = lambda: ops.gemm_nt(a, b, out, bias=bias, epilogue=ops.EPI_RESIDUAL, aux=res)
    elif epi == "gelu":
        out = torch.empty(m, n, device="cuda", dtype=dt)
        pre = torch.empty(m, n, device="cuda", dtype=dt)
        fn = lambda: ops.gemm_nt(a, b, out, bias=bias, epilogue=ops.EPI_GELU, aux_out=pre)
    elif epi == "gelub":
        out = torch.empty(m, n, device="cuda", dtype=dt)
        pre = torch.randn(m, n, device="cuda").to(dt)
        fn = lambda: ops.gemm_nt(a, b, out, epilogue=ops.EPI_GELU_BWD, aux=pre)
    us = timeit(fn)
    fl = 2.0 * m * n * k
    if name != "4096^3":
        tot_us += us; tot_fl += fl
    print(f"NT {name:9s} {m:6d}x{n:5d}x{k:5d} {epi:6s} {us:8.1f} us  {fl / us / 1e6:7.1f} TF")
print(f"NT layer total {tot_us:.1f} us  {tot_fl / tot_us / 1e6:.1f} TF")
tot_us = tot_fl = 0
for name, m, n, t in shapes_tn:
    a = torch.randn(t, m, device="cuda").to(dt)
    b = torch.randn(t, n, device="cuda").to(dt)
    out = torch.empty(m, n, device="cuda", dtype=torch.float32)
    us = timeit(lambda: ops.gemm_tn(a, b, out))
    fl = 2.0 * m * n * t
    if name != "4096^3":
        tot_us += us; tot_fl += fl
    print(f"TN {name:9s} {m:6d}x{n:5d}x{t:5d}        {us:8.1f} us  {fl / us / 1e6:7.1f} TF")
print(f"TN layer total {tot_us:.1f} us  {tot_fl / tot_us / 1e6:.1f} TF")
# the four wgrads of a block as one grouped launch
import ctypes as C
from eoe_amd import _lib
args = (_lib.GemmArgs * 4)()
keep = []
fl = 0
for i, (name, m, n, t) in enumerate(shapes_tn[:4]):
    a = torch.randn(t, m, device="cuda").to(dt)
    b = torch.randn(t, n, device="cuda").to(dt)
    out = torch.empty(m, n, device="cuda", dtype=torch.float32)
    keep += [a, b, out]
    args[i] = _lib.GemmArgs(a.data_ptr(), b.data_ptr(), out.data_ptr(), None, None, None, None, m, n, t, m, n, n, 0,
                            ops.dtype_code(dt), 0, 1, 0, 1.0)
    fl += 2.0 * m * n * t
st = torch.cuda.current_stream().cuda_stream
us = timeit(lambda: _lib.check(_lib.lib.eoe_gemm_tn_grouped(args, 4, st), "g"))
print(f"TN grouped block wgrad (216 tiles)           {us:8.1f} us  {fl / us / 1e6:7.1f} TF")
