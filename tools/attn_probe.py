"""times eoe_attn_fwd / eoe_attn_bwd at the ViT-B/32 shape (256 images x 50 tokens x 12 heads); attn_flags 1 = the one-wave backward kernel"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eoe_amd import ops, _lib


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
for dt in (torch.half, torch.bfloat16):
    qkv = torch.randn(n * L, 3 * D, device="cuda").to(dt)
    do = torch.randn(n * L, D, device="cuda").to(dt)
    out = torch.empty(n * L, D, device="cuda", dtype=dt)
    res = {}
    for rnd in range(3):
        for flags in (1, 0):
            _lib.check(_lib.lib.eoe_set_option(b"attn_flags", flags), "opt")
            dqkv = torch.empty_like(qkv)
            db = torch.zeros(3 * D, device="cuda")
            t0 = timed(lambda: ops.attn_bwd(qkv, do, dqkv, n, L, H))
            t1 = timed(lambda: ops.attn_bwd(qkv, do, dqkv, n, L, H, dbias=db))
            res[flags] = (dqkv.float().clone(), db.clone())
            print(f"{dt} flags {flags}: attn_bwd {t0:.1f} us   + bias sums {t1:.1f} us", flush=True)
    print("   max |d dqkv|", (res[0][0] - res[1][0]).abs().max().item(), " max rel |d dbias|",
          ((res[0][1] - res[1][1]).abs().max() / res[1][1].abs().max()).item())
print(f"attn_fwd {timed(lambda: ops.attn_fwd(qkv, out, n, L, H)):.1f} us")
