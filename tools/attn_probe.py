"""times eoe_attn_fwd / eoe_attn_bwd at the ViT-B/32 shape (256 images x 50 tokens x 12 heads)"""
import torch
from eoe_amd import ops


def timed(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


n, L, H = 256, 50, 12
D = 64 * H
qkv = torch.randn(n * L, 3 * D, device="cuda").half()
do = torch.randn(n * L, D, device="cuda").half()
out = torch.empty(n * L, D, device="cuda", dtype=torch.half)
dqkv = torch.empty_like(qkv)
db = torch.zeros(3 * D, device="cuda")
print(f"attn_fwd {timed(lambda: ops.attn_fwd(qkv, out, n, L, H)):.1f} us   attn_bwd {timed(lambda: ops.attn_bwd(qkv, do, dqkv, n, L, H)):.1f} us   "
      f"attn_bwd + bias sums {timed(lambda: ops.attn_bwd(qkv, do, dqkv, n, L, H, dbias=db)):.1f} us", flush=True)
