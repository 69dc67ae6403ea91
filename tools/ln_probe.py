"""Times eoe_layernorm_bwd (12800 x 768, the ViT-B/32 shape) with and without its column reductions, to price the
atomics at the kernel's tail.  Usage: python tools/ln_probe.py"""
import torch

from eoe_amd import ops


def timed(fn, iters=50):
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


def main():
    rows, D = 12800, 768
    dev = "cuda"
    dy = torch.randn(rows, D, device=dev).half()
    x = torch.randn(rows, D, device=dev)
    stats = torch.stack([x.mean(1), 1 / x.std(1)], 1).contiguous()
    gamma = torch.randn(D, device=dev)
    dres = torch.randn(rows, D, device=dev)
    dx = torch.empty(rows, D, device=dev)
    dx16 = torch.empty(rows, D, device=dev, dtype=torch.half)
    dg, db, ds = (torch.zeros(D, device=dev) for _ in range(3))
    byt = (2 + 4 + 4 + 4 + 2) * rows * D
    for name, kw in (("dx only", {}), ("+dgamma/dbeta", dict(dgamma=dg, dbeta=db)),
                     ("+dgamma/dbeta+dxsum", dict(dgamma=dg, dbeta=db, dxsum=ds))):
        t = timed(lambda: ops.layernorm_bwd(dy, x, stats, gamma, rows, D, D, dx, D, dres=dres, dx16=dx16, **kw))
        print(f"{name:24s} {t:7.1f} us  {byt / t / 1e6:6.2f} TB/s", flush=True)


if __name__ == "__main__":
    main()
