"""helpers for the -m gpu parity tests (HIP path through the C ABI vs the CPU oracle)"""
import numpy as np
import torch

from oracle import fill

DTYPES = [torch.bfloat16, torch.float16]
EPS16 = {torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}   # half-ulp relative rounding error


def t16(name, shape, std, dtype, mean=0.0):
    """deterministic tensor already rounded to the 16-bit dtype; returns (gpu 16-bit, cpu fp32 of the same values)"""
    a = torch.from_numpy(fill.fill(name, shape, std=std, mean=mean)).to(dtype)
    return a.cuda(), a.float()


def f32(name, shape, std, mean=0.0):
    a = torch.from_numpy(fill.fill(name, shape, std=std, mean=mean))
    return a.cuda(), a.clone()


def assert_close(got, ref, rtol, atol, what=""):
    got = got.detach().float().cpu().double()
    ref = ref.detach().double()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert torch.isfinite(got).all(), f"{what}: non-finite values in the result"
    err = (got - ref).abs()
    tol = atol + rtol * ref.abs()
    bad = err > tol
    if bad.any():
        idx = bad.nonzero()[:5].tolist()
        worst = err.argmax().item()
        raise AssertionError(
            f"{what}: {int(bad.sum())}/{bad.numel()} elements out of tolerance (rtol={rtol}, atol={atol}); "
            f"max abs err {err.max().item():.3e} at flat index {worst} (got {got.flatten()[worst].item():.6g}, "
            f"ref {ref.flatten()[worst].item():.6g}); ref rms {ref.pow(2).mean().sqrt().item():.3e}; "
            f"first bad indices {idx}")


def rel_rms(got, ref):
    got = got.detach().float().cpu().double()
    ref = ref.detach().double()
    return ((got - ref).pow(2).mean().sqrt() / (ref.pow(2).mean().sqrt() + 1e-30)).item()


def conditioned_tol(base, sens32, dtype, k=150.0):
    """tolerance for a quantity that two fp32 evaluations (reference vs oracle, or oracle fp32 vs fp64) already disagree on
    by `sens32` (relative): rounding the GEMM operands to 16 bit perturbs the same ill-conditioned sums harder -- measured
    on WideResNet at N=4 (tools/wrn_sens.py) 50-120 x for fp16, so k = 150 (x 8 for bf16's 3 fewer mantissa bits).
    Returns None when the allowance exceeds 50 %: the quantity is cancellation noise at this size and pins nothing."""
    t = max(base, k * (EPS16[dtype] / EPS16[torch.float16]) * sens32)
    return None if t > 0.5 else t
