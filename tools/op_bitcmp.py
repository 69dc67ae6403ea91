"""Bitwise comparison of a few element-wise ops between two builds: `python tools/op_bitcmp.py dump OUT.pt` inside each checkout, then
`python tools/op_bitcmp.py cmp A.pt B.pt`."""
import os, sys
import torch


def dump(path):
    sys.path.insert(0, os.getcwd())
    from eoe_amd import ops
    g = torch.Generator(device="cuda").manual_seed(5)
    rows, D = 12800, 768
    out = {}
    x = torch.randn(rows, D, device="cuda", generator=g) * 1.7 + 0.3
    gamma = torch.randn(D, device="cuda", generator=g)
    beta = torch.randn(D, device="cuda", generator=g)
    for dt in (torch.float16, torch.bfloat16):
        y = torch.empty(rows, D, device="cuda", dtype=dt)
        stats = torch.empty(rows, 2, device="cuda")
        ops.layernorm_fwd(x, gamma, beta, rows, D, D, y, stats)
        out[f"lnf_y_{dt}"] = y.clone(); out[f"lnf_s_{dt}"] = stats.clone()
        dy16 = (torch.randn(rows, D, device="cuda", generator=g) * 0.01).to(dt)
        dres = torch.randn(rows, D, device="cuda", generator=g) * 0.01
        for name, dy, dr in (("16res", dy16, dres), ("16", dy16, None), ("32res", dy16.float(), dres), ("32", dy16.float(), None)):
            dx = torch.empty(rows, D, device="cuda")
            dx16 = torch.empty(rows, D, device="cuda", dtype=dt)
            dg, db, ds = (torch.zeros(D, device="cuda") for _ in range(3))
            ops.layernorm_bwd(dy, x, stats, gamma, rows, D, D, dx, D, dres=dr, dx16=dx16, dgamma=dg, dbeta=db, dxsum=ds)
            for k, v in (("dx", dx), ("dx16", dx16), ("dg", dg), ("db", db), ("ds", ds)):
                out[f"lnb_{name}_{k}_{dt}"] = v.clone()
        w = torch.randn(3072, 768, device="cuda", generator=g)
        d, dtt = ops.cast_transpose(w, dt)
        out[f"ct_d_{dt}"] = d.clone(); out[f"ct_t_{dt}"] = dtt.clone()
    torch.cuda.synchronize()
    torch.save({k: v.cpu() for k, v in out.items()}, path)


def cmp(a, b):
    A, B = torch.load(a), torch.load(b)
    for k in A:
        same = torch.equal(A[k], B[k])
        extra = "" if same else f"  max|d| {(A[k].float() - B[k].float()).abs().max().item():.3e}  n {int((A[k] != B[k]).sum())}"
        print(("same " if same else "DIFF ") + k + extra)


if sys.argv[1] == "dump":
    dump(sys.argv[2])
else:
    cmp(sys.argv[2], sys.argv[3])
