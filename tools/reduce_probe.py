"""times the partial-row finish kernel (multi_reduce_kernel) alone, through the ops that end with it: LayerNorm backward (R = 512 rows x 3 x 768),
attention backward with bias sums (R = 256 x 2304), cold (partials flushed out of the caches by a 1 GB fill) and warm"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eoe_amd import ops, _lib

rows, D, n, L, H = 12800, 768, 256, 50, 12
dt = torch.float16
x = torch.randn(rows, D, device="cuda")
g = torch.randn(D, device="cuda")
stats = torch.rand(rows, 2, device="cuda") + 0.5
dy = (torch.randn(rows, D, device="cuda") * 0.01).to(dt)
dx = torch.empty(rows, D, device="cuda")
dg, db, ds = (torch.zeros(D, device="cuda") for _ in range(3))
qkv = torch.randn(rows, 3 * D, device="cuda").to(dt)
do = torch.randn(rows, D, device="cuda").to(dt)
dqkv = torch.empty_like(qkv)
dbias = torch.zeros(3 * D, device="cuda")
junk = torch.empty(1 << 28, device="cuda")
for cold in (0, 1):
    _lib.prof_enable(True)
    for it in range(12):
        ops.layernorm_bwd(dy, x, stats, g, rows, D, D, dx, D, dgamma=dg, dbeta=db, dxsum=ds)
        if cold:
            junk.fill_(1.0)
        ops.attn_bwd(qkv, do, dqkv, n, L, H, dbias=dbias)
    torch.cuda.synchronize()
    p = _lib.prof_collect()
    _lib.prof_enable(False)
    for k, v in p.items():
        print("cold" if cold else "warm", k, v["launches"], f"{v['total_ms'] * 1e3 / max(1, v['launches']):.1f} us", f"{v['bytes'] / max(1, v['launches']) / 1e6:.1f} MB")
