import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eoe_amd import ops, _lib
from eoe_amd._lib import lib, check
p = lambda t: None if t is None else t.data_ptr()
st = torch.cuda.current_stream().cuda_stream
for (n, H, C, slope) in ((3, 7, 128, 0.0), (2, 5, 64, 0.0), (5, 13, 64, 0.0), (2, 7, 64, 0.0), (70, 1, 64, 0.01), (256, 56, 64, 0.0)):
    torch.manual_seed(0)
    y = torch.randn(n, H, H, C, device="cuda"); dout = torch.randn(n, H, H, C, device="cuda")
    gamma = torch.rand(C, device="cuda") + 0.5; beta = torch.randn(C, device="cuda") * 0.1
    mean = y.double().mean((0, 1, 2)); var = y.double().var((0, 1, 2), unbiased=False); rstd = (var + 1e-5).rsqrt()
    stats = torch.cat([mean, rstd]).float()
    red = torch.empty(ops.BN_SCRATCH * C, device="cuda"); dy = torch.empty(n * H * H, C, device="cuda")
    dg = torch.empty(C, device="cuda"); db = torch.empty(C, device="cuda")
    check(lib.eoe_bn_act_pool_bwd(p(y), p(stats), p(gamma), p(beta), p(dout), p(red), p(dy), 1, p(dg), p(db), n, H, H, C, 1, 0, 1, 0, slope, _lib.EOE_F16, st), "b")
    yd = y.double().requires_grad_(True); gd = gamma.double().requires_grad_(True); bd = beta.double().requires_grad_(True)
    xh = (yd - mean) * rstd
    # treat mean / rstd as functions of y (training-mode BN)
    m2 = yd.mean((0, 1, 2)); v2 = yd.var((0, 1, 2), unbiased=False); xh = (yd - m2) * (v2 + 1e-5).rsqrt()
    z = xh * gd + bd
    out = torch.nn.functional.leaky_relu(z, slope)
    (out * dout.double()).sum().backward()
    rr = lambda a, b: ((a.double() - b).norm() / (b.norm() + 1e-30)).item()
    print((n, H, C, slope), "dgamma", rr(dg, gd.grad), "dbeta", rr(db, bd.grad), "dy", rr(dy.view(n, H, H, C), yd.grad), "nan", torch.isnan(dy).any().item())
