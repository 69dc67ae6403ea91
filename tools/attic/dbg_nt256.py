import os, sys, math
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eoe_amd.ops as ops
from eoe_amd import _lib
torch.manual_seed(0)
_lib.set_option("nt_flags", 1 | 128)
for dt in (torch.float16, torch.bfloat16):
    for (M, N, K) in [(128, 128, 64), (128, 128, 128), (160, 256, 64), (160, 256, 192), (320, 512, 64), (200, 256, 128)]:
        a = torch.randn(M, K, device="cuda").to(dt); b = torch.randn(N, K, device="cuda").to(dt)
        out = torch.empty(M, N, device="cuda", dtype=dt)
        ops.gemm_nt(a, b, out)
        ref = a.double() @ b.double().t()
        err = (out.double() - ref).abs()
        bad = (err > 0.05 * ref.abs() + 0.1).nonzero()
        rows = sorted(set(bad[:, 0].tolist())); cols = sorted(set(bad[:, 1].tolist()))
        print(dt, (M, N, K), "bad", len(bad), "rows", rows[:12], "cols", cols[:20])
        # residual
        out32 = torch.empty(M, N, device="cuda"); res = torch.randn(M, N, device="cuda"); bias = torch.randn(N, device="cuda")
        ops.gemm_nt(a, b, out32, bias=bias, epilogue=ops.EPI_RESIDUAL, aux=res)
        ref2 = ref + bias.double() + res.double()
        err = (out32.double() - ref2).abs()
        bad = (err > 1e-3 * ref2.abs() + 1e-2).nonzero()
        rows = sorted(set(bad[:, 0].tolist())); cols = sorted(set(bad[:, 1].tolist()))
        print("   residual bad", len(bad), "rows", rows[:12], "cols", cols[:20], "max err", err.max().item())
