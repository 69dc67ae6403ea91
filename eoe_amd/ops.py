"""Tensor-level wrappers and torch.autograd.Function bindings over the C ABI of libeoe_hip.so.

PyTorch is used here for device memory, streams and the autograd tape only: every arithmetic operation below
is a hand-written HIP kernel reached through ctypes (eoe_amd._lib).  All functions require CUDA (ROCm) tensors
and raise otherwise -- there is no CPU path in this package.
"""
import ctypes as C
import math
import os
import weakref
from typing import Optional

import torch

from . import _lib
from ._lib import lib, check, GemmArgs, EPI_NONE, EPI_GELU, EPI_RESIDUAL, EPI_GELU_BWD

# ------------------------------------------------------------------------------------------------ config
# fp16 operands (11-bit significand) meet the 1e-3 parity bar against the fp32 reference and are what the
# reference's own GPU CLIP path stores its weights in (clip/model.py:371-392); bf16 is selectable.
_compute_dtype = torch.float16


def set_compute_dtype(dt):
    """16-bit storage / MFMA operand type of activations and weight copies: torch.bfloat16 or torch.float16"""
    global _compute_dtype
    if isinstance(dt, str):
        dt = {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "fp16": torch.float16, "f16": torch.float16,
              "float16": torch.float16}[dt]
    if dt not in (torch.bfloat16, torch.float16):
        raise ValueError(f"unsupported compute dtype {dt}")
    _compute_dtype = dt


# Gradient (loss) scale.  fp16's smallest subnormal is 6e-8 and the backward chain keeps its dY tensors in 16 bits: at the benchmark
# batch (a mean over 256 images, 12 800 token rows) the gradients of late training steps fall below that and are flushed, which the
# 10-step trajectory of the 12-layer ViT shows as a drift to 3e-3 off the fp32 reference (tests/test_gpu_parity_big.py; bf16 has
# fp32's exponent range and is unaffected).  With a scale S the objectives' backward kernels multiply dL/df by S (a power of two:
# exact) and FusedAdam / FusedSGD multiply every gradient by 1/S before using it -- the same update in exact arithmetic, no
# underflow; `p.grad` then holds S times the gradient (`grad_scale()` tells by how much).  Default 1 (off); the trainers and
# bench.py set 256 for fp16.  Only with the eoe_amd optimisers: a stock torch optimiser would see the scaled gradients.
_grad_scale = 1.0


def set_grad_scale(scale: float):
    global _grad_scale
    scale = float(scale)
    if not (scale > 0 and math.log2(scale) == int(math.log2(scale))):
        raise ValueError("the gradient scale must be a positive power of two")
    _grad_scale = scale


def grad_scale() -> float:
    return _grad_scale


def default_grad_scale(dtype=None) -> float:
    """256 for fp16 compute, 1 otherwise"""
    return 256.0 if (dtype or _compute_dtype) == torch.float16 else 1.0


def compute_dtype():
    return _compute_dtype


def dtype_code(dt) -> int:
    if dt == torch.float16:
        return _lib.EOE_F16
    if dt == torch.bfloat16:
        return _lib.EOE_BF16
    raise ValueError(f"not a 16-bit compute dtype: {dt}")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream() -> int:
    """the current stream's handle.  torch.cuda.current_stream() builds a Stream object through four Python layers (10 us; the step asked
    for it 35 times: 0.3 ms of host time per step); the raw getter is one C call"""
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _chk(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("eoe_amd ops need tensors on the GPU (there is no CPU fallback)")


# ------------------------------------------------------------------------------------------------ raw ops
def gemm_nt(a, b, out, bias=None, epilogue=EPI_NONE, aux=None, aux_out=None, accumulate=False, alpha=1.0,
            colsum_out=None, colsum_atomic=False, colstats_ws=None, split_k=False):
    """out[M,N] = a[M,K] @ b[N,K]^T  (+ epilogue); a, b 16-bit row-major (row stride may exceed K).
    colstats_ws: fp32 [ceil(M/64), 2, N] -> receives the per-64-row column sums / sums of squares of the fp32 result"""
    _chk(a, b, out, bias, aux, aux_out, colsum_out)
    if a.shape[1] % 64:               # the MFMA k-tile is 64 deep: zero-pad ragged reductions (CNN28: K = 1568, 32)
        pad = (a.shape[1] + 63) // 64 * 64 - a.shape[1]
        a, b = torch.nn.functional.pad(a, (0, pad)), torch.nn.functional.pad(b, (0, pad))
    M, K = a.shape
    N = b.shape[0]
    assert b.shape[1] == K and out.shape == (M, N), (a.shape, b.shape, out.shape)
    assert a.stride(1) == 1 and b.stride(1) == 1 and out.stride(1) == 1 and a.dtype == b.dtype
    g = GemmArgs(_p(a), _p(b), _p(out), _p(bias), _p(aux), _p(aux_out), _p(colsum_out), M, N, K, a.stride(0), b.stride(0),
                 out.stride(0), aux.stride(0) if aux is not None else 0, dtype_code(a.dtype), epilogue,
                 1 if out.dtype == torch.float32 else 0, 1 if accumulate else 0, float(alpha))
    if colsum_out is not None and not colsum_atomic:      # fused column sums through partial rows (no atomics)
        nbytes = (M + 63) // 64 * N * 4            # EOE_NT_COLSUM_WORKSPACE_BYTES
        g.workspace, g.workspace_bytes = _p(scratch("nt_colsum_ws", (nbytes,), torch.uint8, a.device)), nbytes
    if colstats_ws is not None:
        g.workspace, g.workspace_bytes, g.colstats = _p(colstats_ws), colstats_ws.numel() * 4, 1
    g.split_k = 1 if split_k else 0          # hint (eoe_hip.h): small M behind a long K as k-ranges + an in-order sum
    if split_k or (M >= 2048 and N % 256 == 0):      # split-k partial tiles / the eight-wave kernel's stream-K form (eoe_hip.h, eoe_gemm_args.sk_workspace)
        ws = nt_sk_workspace(a.device)
        g.sk_workspace, g.sk_workspace_bytes = _p(ws), ws.numel()
    check(lib.eoe_gemm_nt(C.byref(g), _stream()), "eoe_gemm_nt")
    return out


TN_WORKSPACE_BYTES = 512 * 256 * 128 * 4      # EOE_TN_WORKSPACE_BYTES of include/eoe_hip.h


def gemm_tn(a, b, out, accumulate=False, alpha=1.0):
    """out[M,N] (fp32) = a[T,M]^T @ b[T,N]; a, b 16-bit row-major"""
    _chk(a, b, out)
    T, M = a.shape
    N = b.shape[1]
    assert b.shape[0] == T and out.shape == (M, N) and out.dtype == torch.float32
    assert a.stride(1) == 1 and b.stride(1) == 1 and out.stride(1) == 1 and a.dtype == b.dtype
    ws = scratch("tn_ws", (TN_WORKSPACE_BYTES,), torch.uint8, a.device)
    g = GemmArgs(_p(a), _p(b), _p(out), None, None, None, None, M, N, T, a.stride(0), b.stride(0), out.stride(0), 0,
                 dtype_code(a.dtype), EPI_NONE, 1, 1 if accumulate else 0, float(alpha), _p(ws), TN_WORKSPACE_BYTES)
    check(lib.eoe_gemm_tn(C.byref(g), _stream()), "eoe_gemm_tn")
    return out


def cast16(src: torch.Tensor, dtype=None, out=None):
    _chk(src)
    dtype = dtype or _compute_dtype
    src = src.contiguous()
    if out is None:
        out = torch.empty(src.shape, dtype=dtype, device=src.device)
    check(lib.eoe_cast(_p(src), _p(out), src.numel(), dtype_code(out.dtype), _stream()), "eoe_cast")
    return out


def cast_transpose(src: torch.Tensor, dtype=None, want=True, want_t=True):
    """fp32 [R,C] -> (16-bit [R,C], 16-bit [C,R])"""
    _chk(src)
    dtype = dtype or _compute_dtype
    src = src.contiguous()
    R, Cc = src.shape
    d = torch.empty((R, Cc), dtype=dtype, device=src.device) if want else None
    dt = torch.empty((Cc, R), dtype=dtype, device=src.device) if want_t else None
    check(lib.eoe_cast_transpose(_p(src), _p(d), _p(dt), R, Cc, dtype_code(dtype), _stream()), "eoe_cast_transpose")
    return d, dt


def cast_colsum(x: torch.Tensor, out16: torch.Tensor, colsum_out: torch.Tensor, accumulate=False):
    """out16 = 16-bit copy of x (fp32 [rows, cols]); colsum_out[c] (+)= sum_r x[r, c]"""
    _chk(x, out16, colsum_out)
    rows, cols = x.shape
    part = scratch("cast_colsum_part", (256 * cols,), torch.float32, x.device)          # EOE_CAST_COLSUM_PARTIALS rows: no atomics
    check(lib.eoe_cast_colsum(_p(x), _p(out16), _p(colsum_out), _p(part), rows, cols, dtype_code(out16.dtype),
                              1 if accumulate else 0, _stream()), "eoe_cast_colsum")
    return out16


def colsum(x16: torch.Tensor, out: torch.Tensor, accumulate=False):
    _chk(x16, out)
    rows, cols = x16.shape
    if not accumulate:      # partial rows + reduce: no atomics, no memset (reproducible; nothing but kernels under graph capture)
        part = scratch("colsum_part", (256 * cols,), torch.float32, x16.device)
        check(lib.eoe_colsum_det(_p(x16), x16.stride(0), _p(out), _p(part), rows, cols, dtype_code(x16.dtype), _stream()),
              "eoe_colsum_det")
        return out
    check(lib.eoe_colsum(_p(x16), x16.stride(0), _p(out), rows, cols, dtype_code(x16.dtype), 1, _stream()), "eoe_colsum")
    return out


def layernorm_fwd(x, gamma, beta, rows, D, ldx, out, stats, eps=1e-5):
    _chk(x, gamma, beta, out, stats)
    out_f32 = 1 if out.dtype == torch.float32 else 0
    code = dtype_code(out.dtype) if not out_f32 else _lib.EOE_BF16
    check(lib.eoe_layernorm_fwd(_p(x), ldx, _p(gamma), _p(beta), _p(out), _p(stats), rows, D, float(eps), code,
                                out_f32, _stream()), "eoe_layernorm_fwd")
    return out


LN_SCRATCH_ROWS = 512          # EOE_LN_PARTIALS (include/eoe_hip.h)


def layernorm_bwd(dy, x, stats, gamma, rows, D, ldx, dx_out, ld_out, dres=None, dx16=None, dgamma=None, dbeta=None,
                  dxsum=None):
    _chk(dy, x, stats, gamma, dx_out, dres, dx16, dgamma, dbeta, dxsum)
    dy_f32 = 1 if dy.dtype == torch.float32 else 0
    code = dtype_code(dy.dtype) if not dy_f32 else (dtype_code(dx16.dtype) if dx16 is not None else _lib.EOE_BF16)
    red = scratch("ln_red", (LN_SCRATCH_ROWS * 3 * D,), torch.float32, dy.device) if (dgamma is not None or dxsum is not None) else None
    check(lib.eoe_layernorm_bwd(_p(dy), dy_f32, _p(x), ldx, _p(stats), _p(gamma), _p(dres), _p(dx_out), ld_out,
                                _p(dx16), _p(dgamma), _p(dbeta), _p(dxsum), _p(red), rows, D, code, _stream()), "eoe_layernorm_bwd")
    return dx_out


def attn_fwd(qkv, out, n, L, heads):
    _chk(qkv, out)
    check(lib.eoe_attn_fwd(_p(qkv), _p(out), n, L, heads, dtype_code(qkv.dtype), _stream()), "eoe_attn_fwd")
    return out


def attn_bwd(qkv, dout, dqkv, n, L, heads, dbias=None):
    """dqkv from (qkv, dout); dbias (optional fp32 [3D]) += column sums of dqkv (the in_proj bias gradient)"""
    _chk(qkv, dout, dqkv, dbias)
    part = scratch("attn_bias_part", (n * 3 * heads * 64,), torch.float32, qkv.device) if dbias is not None else None
    check(lib.eoe_attn_bwd(_p(qkv), _p(dout), _p(dqkv), _p(dbias), _p(part), n, L, heads, dtype_code(qkv.dtype), _stream()),
          "eoe_attn_bwd")
    return dqkv


def patchify(x, patch, mean=None, std=None, dtype=None):
    _chk(x, mean, std)
    dtype = dtype or _compute_dtype
    x = x.contiguous().float()
    n, c, res, res2 = x.shape
    assert c == 3 and res == res2
    g = res // patch
    out = torch.empty((n * g * g, 3 * patch * patch), dtype=dtype, device=x.device)
    check(lib.eoe_patchify(_p(x), _p(mean), _p(std), _p(out), n, res, patch, dtype_code(dtype), _stream()),
          "eoe_patchify")
    return out


# ------------------------------------------------------------------------------------------------ weight copies
class _Shadow:
    """16-bit MFMA operand copies of fp32 master weights ([out,in] and transposed [in,out]), refreshed when the
    parameter's version counter or storage changes (the fused optimiser bumps the counter)."""

    def __init__(self):
        # id(param) -> (weakref(param), tag, copy, transposed copy).  The weak reference guards against a dead
        # model's copies being handed to a new parameter that reuses the address, version and storage.
        self.cache = {}

    def get(self, p: torch.Tensor, want=True, want_t=True, view2d=None):
        key = id(p)
        tag = (p._version, p.data_ptr(), _compute_dtype, want, want_t)
        # while a HIP graph is being captured (eoe_amd.GraphedStep) the cast must be part of the graph -- every replay
        # sees new weights -- and its output is not real until a replay ran: neither read nor fill the cache
        capturing = torch.cuda.is_current_stream_capturing()
        hit = self.cache.get(key)
        if not capturing and hit is not None and hit[0]() is p and hit[1] == tag:
            return hit[2], hit[3]
        src = p.detach()
        if view2d is not None:
            src = src.reshape(view2d)
        d, dt = cast_transpose(src, _compute_dtype, want, want_t)
        if not capturing:
            self.cache[key] = (weakref.ref(p, lambda _r, k=key, c=self.cache: c.pop(k, None)), tag, d, dt)
        return d, dt


    def entry(self, p):
        """(copy, transposed copy) of a 2-D parameter if both exist and are current, else None: the pair `eoe_adam_tiles` may rewrite in place"""
        hit = self.cache.get(id(p))
        if hit is None or hit[0]() is not p or hit[2] is None or hit[3] is None:
            return None
        if hit[1] != (p._version, p.data_ptr(), _compute_dtype, True, True) or p.dim() != 2 or hit[2].shape != p.shape:
            return None
        return hit[2], hit[3]

    def mark(self, p, d, dt):
        """the optimiser has just written `d` / `dt` from the updated parameter (and bumped its version): they are current"""
        key = id(p)
        self.cache[key] = (weakref.ref(p, lambda _r, k=key, c=self.cache: c.pop(k, None)), (p._version, p.data_ptr(), _compute_dtype, True, True), d, dt)

    def refresh(self, params):
        """bring the [out,in] + transposed copies of all (2-D, contiguous) `params` up to date with ONE batched launch
        (`eoe_cast_transpose_multi`); the per-use `get` calls of the step then hit the cache.  No-op under graph capture."""
        if torch.cuda.is_current_stream_capturing():
            return
        jobs, entries = [], []
        for p in params:
            tag = (p._version, p.data_ptr(), _compute_dtype, True, True)
            hit = self.cache.get(id(p))
            if hit is not None and hit[0]() is p and hit[1] == tag:
                continue
            src = p.detach()
            if src.dim() != 2 or not src.is_contiguous() or src.shape[0] % 4 or src.shape[1] % 4 or not src.is_cuda:
                continue                                    # left to the per-use path
            R, Cc = src.shape
            d = torch.empty((R, Cc), dtype=_compute_dtype, device=src.device)
            dt = torch.empty((Cc, R), dtype=_compute_dtype, device=src.device)
            jobs.append(_lib.CastJob(_p(src), _p(d), _p(dt), R, Cc))
            entries.append((p, tag, d, dt))
        if not jobs:
            return
        arr = (_lib.CastJob * len(jobs))(*jobs)
        check(lib.eoe_cast_transpose_multi(arr, len(jobs), dtype_code(_compute_dtype), _stream()), "eoe_cast_transpose_multi")
        for p, tag, d, dt in entries:
            key = id(p)
            self.cache[key] = (weakref.ref(p, lambda _r, k=key, c=self.cache: c.pop(k, None)), tag, d, dt)


shadow = _Shadow()

_scratch = {}


def scratch(name, shape, dtype, device):
    """persistent scratch buffers reused across calls, one set per (device, stream): work on one stream is sequential, so one
    buffer per shape is enough there, while two streams (or two models driven from two streams) never share one"""
    key = (name, tuple(shape), dtype, device, _stream())
    t = _scratch.get(key)
    if t is None:
        t = torch.empty(shape, dtype=dtype, device=device)
        if not torch.cuda.is_current_stream_capturing():     # a buffer from a graph's private pool dies with the graph
            _scratch[key] = t
    return t


def nt_sk_workspace(device):
    """the stream-K workspace of the NT GEMMs (EOE_NT_STREAMK_WORKSPACE_BYTES of include/eoe_hip.h): one per (device, stream), zeroed once --
    every launch leaves its ticket / flag words zeroed again"""
    key = ("nt_sk_ws", device, _stream())
    t = _scratch.get(key)
    if t is None:
        cus = torch.cuda.get_device_properties(device).multi_processor_count
        t = torch.zeros(8192 + cus * 2 * 256 * 256 * 4, dtype=torch.uint8, device=device)
        if not torch.cuda.is_current_stream_capturing():
            _scratch[key] = t
    return t


# id(first parameter of a fused block) -> callable invoked right after that block's backward kernels were enqueued
# (eoe_amd.parallel.GradArena uses it to start the bucket's all-reduce while backward continues)
grad_ready_hooks = {}


def zero_multi(*tensors):
    """zero up to 8 fp32 buffers in ONE launch (instead of one torch fill each)"""
    ts = [t for t in tensors if t is not None and t.numel() > 0]
    for i in range(0, len(ts), 8):
        chunk = ts[i:i + 8]
        for t in chunk:
            assert t.dtype == torch.float32 and t.is_contiguous() and t.is_cuda
        ptrs = (C.c_void_p * len(chunk))(*[t.data_ptr() for t in chunk])
        cnts = (C.c_int * len(chunk))(*[t.numel() for t in chunk])
        check(lib.eoe_zero_multi(ptrs, cnts, len(chunk), _stream()), "eoe_zero_multi")


def _grad_target(p: torch.Tensor):
    """where a parameter gradient is written: a fresh alias of the parameter's registered arena view if it has one
    and is free (p.grad is None) -- a new tensor object over the same memory, so that autograd's AccumulateGrad
    adopts it without a copy -- else a fresh tensor"""
    buf = getattr(p, "_eoe_grad_buf", None)
    if buf is not None and p.grad is None:
        return buf.detach()
    return torch.empty_like(p, memory_format=torch.contiguous_format)


# ------------------------------------------------------------------------------------------------ autograd: linear
class LinearFunction(torch.autograd.Function):
    """y = x @ W^T + b with fp32 in/out, 16-bit MFMA operands (custom_base.py:25-26,47-48 final_linear)"""

    @staticmethod
    def forward(ctx, x, weight, bias):
        _chk(x, weight, bias)
        x2 = x.reshape(-1, x.shape[-1])
        x16 = x2 if x2.dtype == _compute_dtype else cast16(x2.float())
        w16, _ = shadow.get(weight, True, True)
        out = torch.empty((x2.shape[0], weight.shape[0]), dtype=torch.float32, device=x.device)
        gemm_nt(x16, w16, out, bias=bias)
        ctx.save_for_backward(x16, weight, bias)
        ctx.in_shape = x.shape
        ctx.x_dtype = x.dtype
        return out.reshape(*x.shape[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x16, weight, bias = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1]).contiguous().float()
        d16 = cast16(dy2)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            _, w16t = shadow.get(weight, True, True)
            dx = torch.empty((d16.shape[0], weight.shape[1]), dtype=torch.float32, device=dy.device)
            gemm_nt(d16, w16t, dx)
            dx = dx.reshape(ctx.in_shape).to(ctx.x_dtype)
        if ctx.needs_input_grad[1]:
            dw = _grad_target(weight)
            gemm_tn(d16, x16, dw)
        if bias is not None and ctx.needs_input_grad[2]:
            db = _grad_target(bias)
            colsum(d16, db)
        return dx, dw, db


class LinearSmallFunction(torch.autograd.Function):
    """narrow head (N <= 8 outputs, e.g. final_linear(512 -> 1) of CustomNet(clf=True)): exact fp32 kernels"""

    @staticmethod
    def forward(ctx, x, weight, bias):
        _chk(x, weight, bias)
        x2 = x.reshape(-1, x.shape[-1]).contiguous().float()
        w = weight.contiguous()
        M, K = x2.shape
        N = w.shape[0]
        y = torch.empty((M, N), dtype=torch.float32, device=x.device)
        check(lib.eoe_linear_small_fwd(_p(x2), _p(w), _p(bias), _p(y), M, N, K, _stream()), "eoe_linear_small_fwd")
        ctx.save_for_backward(x2, weight, bias)
        ctx.in_shape = x.shape
        return y.reshape(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, weight, bias = ctx.saved_tensors
        M, K = x2.shape
        N = weight.shape[0]
        dy2 = dy.reshape(M, N).contiguous().float()
        dx = torch.empty_like(x2) if ctx.needs_input_grad[0] else None
        dw = _grad_target(weight) if ctx.needs_input_grad[1] else None
        db = _grad_target(bias) if (bias is not None and ctx.needs_input_grad[2]) else None
        if db is not None and dw is None:
            dw = torch.empty_like(weight)
        check(lib.eoe_linear_small_bwd(_p(x2), _p(weight.contiguous()), _p(dy2), _p(dx), _p(dw), _p(db), M, N, K, 0,
                                       _stream()), "eoe_linear_small_bwd")
        return (dx.reshape(ctx.in_shape) if dx is not None else None), (dw if ctx.needs_input_grad[1] else None), db


def linear(x, weight, bias=None):
    if weight.shape[0] <= 8:
        return LinearSmallFunction.apply(x, weight, bias)          # exact fp32 already
    if _parity:
        return LinearParityFunction.apply(x, weight, bias)
    return LinearFunction.apply(x, weight, bias)


# ------------------------------------------------------------------------------------------------ autograd: ViT
class VitEmbedFunction(torch.autograd.Function):
    """images -> residual stream entering block 0 (clip/model.py:220-225): [normalise +] patchify, conv1 as a
    GEMM, class/positional embedding, ln_pre.  Output fp32 [n*L, D]."""

    @staticmethod
    def forward(ctx, x, conv_w, cls, pos, g, b, patch, mean, std):
        _chk(x, conv_w, cls, pos, g, b)
        n = x.shape[0]
        D = conv_w.shape[0]
        L = pos.shape[0]
        patches = patchify(x, patch, mean, std)
        w16, _ = shadow.get(conv_w, True, False, view2d=(D, -1))
        tok = torch.empty((patches.shape[0], D), dtype=torch.float32, device=x.device)
        gemm_nt(patches, w16, tok)
        x0 = torch.empty((n * L, D), dtype=torch.float32, device=x.device)
        y = torch.empty_like(x0)
        stats = torch.empty((n * L, 2), dtype=torch.float32, device=x.device)
        check(lib.eoe_embed_lnpre_fwd(_p(tok), _p(cls), _p(pos), _p(g), _p(b), _p(x0), _p(y), _p(stats), n, L, D, 1e-5,
                                      _stream()), "eoe_embed_lnpre_fwd")
        ctx.save_for_backward(patches, x0, stats, conv_w, cls, pos, g, b)
        ctx.dims = (n, L, D)
        return y

    @staticmethod
    def backward(ctx, dy):
        patches, x0, stats, conv_w, cls, pos, g, b = ctx.saved_tensors
        n, L, D = ctx.dims
        dy = dy.contiguous()
        dtok = torch.empty((n * (L - 1), D), dtype=patches.dtype, device=dy.device)
        dcls, dpos, dg, db = (_grad_target(t) for t in (cls, pos, g, b))
        zero_multi(dcls, dpos, dg, db)
        part = scratch("embed_ln_part", (L * 2 * D,), torch.float32, dy.device)          # ln_pre parameter gradients without atomics
        check(lib.eoe_embed_lnpre_bwd(_p(dy), _p(x0), _p(stats), _p(g), _p(dtok), _p(dcls), _p(dpos), _p(dg), _p(db), _p(part),
                                      n, L, D, dtype_code(patches.dtype), _stream()), "eoe_embed_lnpre_bwd")
        dw = _grad_target(conv_w)
        gemm_tn(dtok, patches, dw.view(D, -1))
        return None, dw, dcls, dpos, dg, db, None, None, None


_BLOCK_PARAMS = ("ln1_g", "ln1_b", "w_in", "b_in", "w_out", "b_out", "ln2_g", "ln2_b", "w_fc", "b_fc", "w_proj", "b_proj")


def _block_ws(M, D, device, dtype):
    """one allocation holding every activation a block saves for backward"""
    es = 2
    sizes = [("xn1", M * D * es), ("qkv", M * 3 * D * es), ("att", M * D * es), ("xn2", M * D * es),
             ("hpre", M * 4 * D * es), ("hact", M * 4 * D * es), ("stats1", M * 2 * 4), ("stats2", M * 2 * 4),
             ("x_mid", M * D * 4)]
    offs, tot = {}, 0
    for k, s in sizes:
        offs[k] = tot
        tot += (s + 255) // 256 * 256
    buf = torch.empty(tot, dtype=torch.uint8, device=device)
    base = buf.data_ptr()
    return buf, {k: base + o for k, o in offs.items()}


VIT_HANDOVER = os.environ.get("EOE_VIT_HANDOVER", "1") != "0"     # block-to-block hand-over of dY(c_proj) in the backward sweep (0: A/B)
VIT_ASYNC_WGRAD = os.environ.get("EOE_VIT_ASYNC_WGRAD", "1") != "0"      # a block's weight gradients on the side stream, under the next block (0: A/B)
# (round 5) the blocks' finish reductions (bias and LayerNorm-parameter gradients out of their partial rows) collected in ONE table and launched
# once at the end of the backward sweep instead of one 17-us kernel per block (0: per block, A/B).  Not with a data-parallel bucket hook on the
# block: its all-reduce reads those gradients right behind the block
VIT_DEFER_FINISH = os.environ.get("EOE_VIT_DEFER_FINISH", "1") != "0"
# 1 (default): every block call orders the stream behind the previous block's weight-gradient launch (two sets of scratch); 2: behind the launch
# before that one (three sets; eoe_hip.h, async_wgrad = 2).  Built in round 5 against the ~12-us stalls the r4 kernel trace shows at the fork
# points; measured, it moves nothing (10.392 / 10.377 against 10.398 / 10.403 ms per step, two interleaved pairs on one box: the stalls are the
# profiler's), so the default stays at 1 and the third set of buffers (180 MB) is not taken
VIT_ASYNC_LAG = int(os.environ.get("EOE_VIT_ASYNC_LAG", "1"))
_vit_red_table = _lib.RedTable()
_vit_args_cache = {}          # id(w_in) -> (signature, the static part of the block's eoe_vit_block_fwd_args as bytes)
_vit_red_seq = 0
_vit_red_stream = None        # the stream the current sweep's blocks were enqueued on
VIT_RED_POOL = 16
_vit_handoff = None
_vit_parity = 0
_vit_pending = None
_vit_deferred_hook = None


def vit_flush_finish():
    """launches the finish reductions the blocks of this backward sweep left in the table (VIT_DEFER_FINISH); called by the autograd engine
    at the end of the pass, harmless at any other time"""
    global _vit_red_seq
    _vit_red_seq = 0
    if _vit_red_table.count:
        # on the stream the sweep's kernels ran on (the engine runs this callback in the thread that called backward(), whose current stream may
        # be another one; the engine orders that stream behind the sweep's only AFTER the callbacks)
        st = _vit_red_stream if _vit_red_stream is not None else _stream()
        check(lib.eoe_red_table_flush(C.byref(_vit_red_table), st), "eoe_red_table_flush")


def vit_side_join():
    """orders the current stream behind the last asynchronous weight-gradient launch and releases what it was reading; called by the
    autograd engine at the end of a backward pass that used the asynchronous path (and harmless at any other time)"""
    global _vit_pending, _vit_deferred_hook
    vit_flush_finish()
    check(lib.eoe_vit_side_join(_stream()), "eoe_vit_side_join")
    _vit_pending = None
    if _vit_deferred_hook is not None:                     # the last block of the sweep: its bucket goes out behind the join
        prev, _vit_deferred_hook = _vit_deferred_hook, None
        prev()



# the last block of the vision tower computes only what its single reader -- ln_post on the class token (clip/model.py:231-232) -- uses
# (EOE_VIT_CLS_ONLY=0, or this flag, restores the full block: A/B and tests)
VIT_CLS_ONLY_LAST = os.environ.get("EOE_VIT_CLS_ONLY", "1") != "0"


class VitBlockFunction(torch.autograd.Function):
    """one ResidualAttentionBlock (clip/model.py:167-188) on the batch-major token matrix, fp32 residual stream
    [n*L, D] in and out; a single C call launches the whole kernel chain."""

    @staticmethod
    def forward(ctx, x, n, heads, ln1_g, ln1_b, w_in, b_in, w_out, b_out, ln2_g, ln2_b, w_fc, b_fc, w_proj, b_proj, cls_only=False):
        _chk(x)
        x = x.contiguous()
        M, D = x.shape
        L = M // n
        ws, ptr = _block_ws(M, D, x.device, _compute_dtype)
        # cls_only (the last block of the tower: eoe_hip.h, eoe_vit_block_fwd_args.cls_only): the output is the [n, D] matrix of class-token
        # rows -- all the head reads --, and the out-projection, LayerNorm-2 and the MLP run on those rows only
        cls_only = bool(cls_only) and L >= 4
        x_out = x.new_empty((n, D)) if cls_only else torch.empty_like(x)
        need_t = torch.is_grad_enabled()
        sh = {k: shadow.get(w, True, True) for k, w in (("in", w_in), ("out", w_out), ("fc", w_fc), ("proj", w_proj))}
        # the argument block of a given block changes between steps only in its activation pointers: the parameter / weight-copy / workspace
        # part is filled once per (block, pointers) and copied (round 5: ~40 ctypes field stores per block and step were 0.4 ms of host time)
        sk_ws = nt_sk_workspace(x.device)
        code = dtype_code(_compute_dtype)
        sig = (n, L, D, heads, code, _p(ln1_g), _p(ln1_b), _p(ln2_g), _p(ln2_b), _p(b_in), _p(b_out), _p(b_fc), _p(b_proj),
               _p(sh["in"][0]), _p(sh["out"][0]), _p(sh["fc"][0]), _p(sh["proj"][0]), _p(sh["in"][1]), _p(sh["out"][1]), _p(sh["fc"][1]),
               _p(sh["proj"][1]), _p(sk_ws))
        hit = _vit_args_cache.get(id(w_in))
        if hit is not None and hit[0] == sig:
            a = _lib.VitBlockFwdArgs.from_buffer_copy(hit[1])
        else:
            a = _lib.VitBlockFwdArgs()
            a.n, a.L, a.D, a.heads, a.dtype, a.eps = n, L, D, heads, code, 1e-5
            a.ln1_g, a.ln1_b, a.ln2_g, a.ln2_b = _p(ln1_g), _p(ln1_b), _p(ln2_g), _p(ln2_b)
            a.b_in, a.b_out, a.b_fc, a.b_proj = _p(b_in), _p(b_out), _p(b_fc), _p(b_proj)
            a.w_in, a.w_out, a.w_fc, a.w_proj = (_p(sh[k][0]) for k in ("in", "out", "fc", "proj"))
            a.w_in_t, a.w_out_t, a.w_fc_t, a.w_proj_t = (_p(sh[k][1]) for k in ("in", "out", "fc", "proj"))
            a.nt_sk_workspace, a.nt_sk_workspace_bytes = _p(sk_ws), sk_ws.numel()
            if len(_vit_args_cache) > 256:
                _vit_args_cache.clear()
            _vit_args_cache[id(w_in)] = (sig, bytes(a))
        a.x_in, a.x_mid, a.x_out = _p(x), ptr["x_mid"], _p(x_out)
        a.xn1, a.qkv, a.att, a.xn2, a.hpre, a.hact = (ptr[k] for k in ("xn1", "qkv", "att", "xn2", "hpre", "hact"))
        a.stats1, a.stats2 = ptr["stats1"], ptr["stats2"]
        a.cls_only = 1 if cls_only else 0
        if _vit_red_table.count:                 # jobs of a backward pass that died before its end-of-pass flush: their scratch is about to be reused
            _vit_red_table.count = 0
        keep = any(ctx.needs_input_grad)            # (grad mode itself is always off inside a Function's forward)
        if not keep:
            a.hpre = None          # forward only (frozen encoder, scoring): the MLP's pre-activation is not kept (79 MB per block)
        check(lib.eoe_vit_block_fwd(C.byref(a), _stream()), "eoe_vit_block_fwd")
        ctx.save_for_backward(x, ws, ln1_g, ln1_b, w_in, b_in, w_out, b_out, ln2_g, ln2_b, w_fc, b_fc, w_proj, b_proj)
        ctx.args = a
        ctx.shadows = sh          # keep the 16-bit copies used by this forward alive until backward
        return x_out

    @staticmethod
    def backward(ctx, dx_out):
        saved = ctx.saved_tensors
        x, ws = saved[0], saved[1]
        params = dict(zip(_BLOCK_PARAMS, saved[2:]))
        M, D = x.shape
        dev = x.device
        dt = torch.float16 if ctx.args.dtype == _lib.EOE_F16 else torch.bfloat16
        dx_out = dx_out.contiguous()
        dx_in = torch.empty_like(x)
        grads = {k: _grad_target(p) for k, p in params.items()}
        b = _lib.VitBlockBwdArgs()
        b.f = ctx.args
        b.dx_out, b.dx_in = _p(dx_out), _p(dx_in)
        b.g_ln1_g, b.g_ln1_b, b.g_ln2_g, b.g_ln2_b = (_p(grads[k]) for k in ("ln1_g", "ln1_b", "ln2_g", "ln2_b"))
        b.g_b_in, b.g_b_out, b.g_b_fc, b.g_b_proj = (_p(grads[k]) for k in ("b_in", "b_out", "b_fc", "b_proj"))
        b.g_w_in, b.g_w_out, b.g_w_fc, b.g_w_proj = (_p(grads[k]) for k in ("w_in", "w_out", "w_fc", "w_proj"))
        b.accumulate = 0
        # hand-over between consecutive blocks of the backward sweep (eoe_hip.h, eoe_vit_block_bwd_args): this block's LayerNorm-1 backward
        # also writes the 16-bit copy of dx_in and leaves its column sums in its partial rows -- what the next block to run would compute
        # with a pass of its own (eoe_cast_colsum).  Two alternating sets of (copy, reduction scratch): the previous call's are still read.
        global _vit_handoff, _vit_parity
        par = _vit_parity = (_vit_parity + 1) % (3 if VIT_ASYNC_LAG >= 2 else 2)      # (async_wgrad = 2: a launch's buffers rest until two later calls have returned)
        h = _vit_handoff
        _vit_handoff = None
        d16_next = scratch(f"d16_next{par}", (M, D), dt, dev)
        b.next_d16 = _p(d16_next) if VIT_HANDOVER else None
        if (h is not None and h["dx"] is dx_out and dx_out._version == h["version"] and h["shape"] == (M, D, ctx.args.n) and h["dt"] == dt
                and h["stream"] == _stream()):
            b.in_d16, b.in_red_scratch = _p(h["d16"]), _p(h["red"])
        # asynchronous weight gradients (eoe_hip.h, eoe_vit_block_bwd_args.async_wgrad): the block's grouped wgrad launch goes to the
        # library's side stream and runs under the NEXT block's kernels.  What it reads -- dh, dqkv, d16_c, the dY of c_proj and the saved
        # activations in `ws` -- must outlive this call: two alternating sets of those scratch buffers (`par`), `ws` parked in `_vit_pending`
        # until the next block's call has returned (it orders the stream behind this launch), and one join when the backward pass ends.
        hook = grad_ready_hooks.get(id(params["ln1_g"]))
        has_hook = hook is not None and hook[0]() is params["ln1_g"]
        # (a data-parallel bucket hook of this block wants its weight gradients: with the asynchronous launch it is fired one block later --
        #  after the next block's call, which orders the stream behind this block's launch -- or by the join at the end of the pass)
        use_async = VIT_ASYNC_WGRAD and not torch.cuda.is_current_stream_capturing()
        # 2: this call waits for the launch before the previous one only (eoe_hip.h); a data-parallel bucket hook reads the previous block's weight
        # gradients right behind this call, so with one installed the call orders the stream behind the previous launch as before (1)
        b.async_wgrad = (1 if has_hook or VIT_ASYNC_LAG == 1 else 2) if use_async else 0
        b.d16_a = _p(scratch(f"d16_a{par}", (M, D), dt, dev))
        b.d16_b = _p(scratch("d16_b", (M, D), dt, dev))
        b.d16_c = _p(scratch(f"d16_c{par}", (M, D), dt, dev))
        b.dh = _p(scratch(f"dh{par}", (M, 4 * D), dt, dev))
        b.dqkv = _p(scratch(f"dqkv{par}", (M, 3 * D), dt, dev))
        b.dx_mid = _p(scratch("dx_mid", (M, D), torch.float32, dev))
        nred = (M + 63) // 64 * 4 * D + 2 * LN_SCRATCH_ROWS * 3 * D + ctx.args.n * 3 * D + 256 * D          # EOE_VIT_RED_SCRATCH(n, L, D)
        # deferred finish: every block of the sweep keeps its own partial rows until the one flush at the end of the pass
        global _vit_red_seq
        defer = VIT_DEFER_FINISH and not has_hook and not torch.cuda.is_current_stream_capturing()
        if defer:
            global _vit_red_stream
            _vit_red_stream = _stream()
            red = scratch(f"vit_red_seq{_vit_red_seq % VIT_RED_POOL}", (nred,), torch.float32, dev)
            _vit_red_seq += 1
            if _vit_red_seq >= VIT_RED_POOL:               # a deeper tower than the pool: launch what is queued before a scratch comes round again
                vit_flush_finish()
            b.red_table = C.pointer(_vit_red_table)
        else:
            red = scratch(f"vit_red{par}", (nred,), torch.float32, dev)
        b.red_scratch = _p(red)
        sk_bytes = torch.cuda.get_device_properties(dev).multi_processor_count * (256 * 256 * 4)      # EOE_TN_STREAMK_WORKSPACE_BYTES
        b.tn_workspace, b.tn_workspace_bytes = _p(scratch("tn_streamk", (sk_bytes,), torch.uint8, dev)), sk_bytes
        check(lib.eoe_vit_block_bwd(C.byref(b), _stream()), "eoe_vit_block_bwd")
        if VIT_HANDOVER:
            _vit_handoff = dict(dx=dx_in, version=dx_in._version, shape=(M, D, ctx.args.n), dt=dt, stream=_stream(), d16=d16_next, red=red)
        global _vit_pending, _vit_deferred_hook
        # what the launches still in flight read: this call's and the previous call's (the call before that has been ordered behind by now)
        _vit_pending = ((x, ws), _vit_pending[0] if _vit_pending else None) if use_async else None
        if _vit_deferred_hook is not None:                 # the previous block's bucket: its weight gradients are complete in stream order now
            prev, _vit_deferred_hook = _vit_deferred_hook, None
            prev()
        if use_async:
            # runs when this backward pass is complete; queued by every block (idempotent): a flag "already queued" would survive a pass
            # that died with an exception and leave the next pass without its join
            torch.autograd.Variable._execution_engine.queue_callback(vit_side_join)
        elif defer:
            torch.autograd.Variable._execution_engine.queue_callback(vit_flush_finish)
        if has_hook:      # (weakref to the parameter, callable): the bucket's all-reduce reads this block's weight gradients
            if use_async:
                _vit_deferred_hook = hook[1]
            else:
                hook[1]()
        return (dx_in, None, None) + tuple(grads[k] for k in _BLOCK_PARAMS) + (None,)


class VitHeadFunction(torch.autograd.Function):
    """ln_post on the class token + projection (clip/model.py:231-234): [n*L, D] fp32 -> [n, out] fp32"""

    @staticmethod
    def forward(ctx, x, n, g, b, proj):
        _chk(x, g, b, proj)
        x = x.contiguous()
        M, D = x.shape
        L = M // n
        cls16 = torch.empty((n, D), dtype=_compute_dtype, device=x.device)
        stats = torch.empty((n, 2), dtype=torch.float32, device=x.device)
        layernorm_fwd(x, g, b, n, D, L * D, cls16, stats)
        if proj is None:
            raise RuntimeError("VisualTransformer without proj is not supported")
        p16, p16t = shadow.get(proj, True, True)          # proj [D, out]; p16t [out, D]
        out = torch.empty((n, proj.shape[1]), dtype=torch.float32, device=x.device)
        gemm_nt(cls16, p16t, out)
        ctx.save_for_backward(x, stats, cls16, g, b, proj)
        ctx.dims = (n, L, D)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, stats, cls16, g, b, proj = ctx.saved_tensors
        n, L, D = ctx.dims
        d16 = cast16(dout.contiguous().float())
        p16, _ = shadow.get(proj, True, True)
        dcls = torch.empty((n, D), dtype=d16.dtype, device=x.device)
        gemm_nt(d16, p16, dcls)                            # [n, out] @ proj[D, out]^T
        dproj = _grad_target(proj)
        gemm_tn(cls16, d16, dproj)                         # [D, out] = cls16^T d16
        dx = torch.empty_like(x)
        dg, db = _grad_target(g), _grad_target(b)
        zero_multi(dx, dg, db)                            # one launch (the class-token rows of dx are written below, the rest stays 0)
        layernorm_bwd(dcls, x, stats, g, n, D, L * D, dx, L * D, dgamma=dg, dbeta=db)
        return dx, None, dg, db, dproj


# ------------------------------------------------------------------------------------------------ autograd: objectives
class HscLossFunction(torch.autograd.Function):
    """HSCTrainer.loss (hsc.py:17-21) as one fused head; inv_count = 1/N (or 1/global N under data parallel)"""

    @staticmethod
    def forward(ctx, feats, labels, nominal_label, inv_count):
        _chk(feats, labels)
        f = feats.contiguous().float()
        n, d = f.shape
        labels = labels.contiguous().to(torch.int64)
        loss = torch.empty(1, dtype=torch.float32, device=f.device)
        losses = torch.empty(n, dtype=torch.float32, device=f.device)
        inv = float(inv_count) if inv_count is not None else 1.0 / n
        check(lib.eoe_hsc_fwd(_p(f), _p(labels), int(nominal_label), _p(loss), None, None, _p(losses), n, d, inv,
                              _stream()), "eoe_hsc_fwd")
        ctx.save_for_backward(f, labels)
        ctx.cfg = (int(nominal_label), inv)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gout):
        f, labels = ctx.saved_tensors
        nominal, inv = ctx.cfg
        n, d = f.shape
        df = torch.empty_like(f)
        gs = gout.contiguous().float().reshape(1)
        check(lib.eoe_hsc_bwd(_p(f), _p(labels), nominal, _p(gs), _p(df), None, n, d, inv * _grad_scale, _lib.EOE_BF16, _stream()),
              "eoe_hsc_bwd")
        return df, None, None, None


def hsc_loss(feats, labels, nominal_label=0, inv_count=None):
    return HscLossFunction.apply(feats, labels, nominal_label, inv_count)


def hsc_score(feats):
    """HSCTrainer.compute_anomaly_score (hsc.py:12-15)"""
    _chk(feats)
    f = feats.detach().contiguous().float()
    n, d = f.shape
    out = torch.empty(n, dtype=torch.float32, device=f.device)
    check(lib.eoe_hsc_score(_p(f), _p(out), n, d, _stream()), "eoe_hsc_score")
    return out


class BceLossFunction(torch.autograd.Function):
    """BCETrainer.loss (bce.py:19-20)"""

    @staticmethod
    def forward(ctx, feats, labels, inv_count):
        _chk(feats, labels)
        x = feats.contiguous().float().reshape(-1)
        n = x.shape[0]
        labels = labels.contiguous().to(torch.int64)
        loss = torch.empty(1, dtype=torch.float32, device=x.device)
        losses = torch.empty(n, dtype=torch.float32, device=x.device)
        inv = float(inv_count) if inv_count is not None else 1.0 / n
        check(lib.eoe_bce_fwd(_p(x), _p(labels), 0, _p(loss), None, _p(losses), n, inv, _stream()), "eoe_bce_fwd")
        ctx.save_for_backward(x, labels)
        ctx.inv = inv
        ctx.shape = feats.shape
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gout):
        x, labels = ctx.saved_tensors
        dx = torch.empty_like(x)
        gs = gout.contiguous().float().reshape(1)
        check(lib.eoe_bce_bwd(_p(x), _p(labels), _p(gs), _p(dx), x.shape[0], ctx.inv * _grad_scale, _stream()), "eoe_bce_bwd")
        return dx.reshape(ctx.shape), None, None


def bce_loss(feats, labels, inv_count=None):
    return BceLossFunction.apply(feats, labels, inv_count)


def bce_score(feats, nominal_label=0):
    """BCETrainer.compute_anomaly_score (bce.py:15-17)"""
    _chk(feats)
    x = feats.detach().contiguous().float().reshape(-1)
    n = x.shape[0]
    out = torch.empty(n, dtype=torch.float32, device=x.device)
    dummy = torch.zeros(n, dtype=torch.int64, device=x.device)
    check(lib.eoe_bce_fwd(_p(x), _p(dummy), int(nominal_label), None, _p(out), None, n, 1.0, _stream()), "eoe_bce_fwd")
    return out


# ------------------------------------------------------------------------------------------------ autograd: other objectives
class ClipLossFunction(torch.autograd.Function):
    """ADClipTrainer.loss (training/clip.py:81-103): -mean log_softmax(100 f/|f| T^T)[pick]"""

    @staticmethod
    def forward(ctx, feats, labels, text, nominal_label, leave_one_out, inv_count):
        _chk(feats, labels, text)
        f = feats.contiguous().float()
        n, d = f.shape
        t = text.detach().contiguous().float()
        assert t.dim() == 2 and t.shape[1] == d and 2 <= t.shape[0] <= 64, t.shape
        labels = labels.contiguous().to(torch.int64)
        loss = torch.empty(1, dtype=torch.float32, device=f.device)
        losses = torch.empty(n, dtype=torch.float32, device=f.device)
        inv = float(inv_count) if inv_count is not None else 1.0 / n
        check(lib.eoe_clip_fwd(_p(f), _p(t), _p(labels), int(nominal_label), 1 if leave_one_out else 0, _p(loss), None, _p(losses),
                               n, d, t.shape[0], inv, _stream()), "eoe_clip_fwd")
        ctx.save_for_backward(f, t, labels)
        ctx.cfg = (int(nominal_label), 1 if leave_one_out else 0, inv)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gout):
        f, t, labels = ctx.saved_tensors
        nominal, loo, inv = ctx.cfg
        df = torch.empty_like(f)
        gs = gout.contiguous().float().reshape(1)
        check(lib.eoe_clip_bwd(_p(f), _p(t), _p(labels), nominal, loo, _p(gs), _p(df), f.shape[0], f.shape[1], t.shape[0],
                               inv * _grad_scale, _stream()), "eoe_clip_bwd")
        return df, None, None, None, None, None


def clip_loss(feats, labels, text, nominal_label=0, leave_one_out=False, inv_count=None):
    return ClipLossFunction.apply(feats, labels, text, nominal_label, leave_one_out, inv_count)


def clip_score(feats, text):
    """ADClipTrainer.compute_anomaly_score (clip.py:66-79): softmax(100 f/|f| (T/|T|)^T)[:, -1]"""
    _chk(feats, text)
    f = feats.detach().contiguous().float()
    t = text.detach().float()
    t = (t / t.norm(dim=-1, keepdim=True)).contiguous()           # T x d, tiny: clip.py:69
    out = torch.empty(f.shape[0], dtype=torch.float32, device=f.device)
    check(lib.eoe_clip_score(_p(f), _p(t), _p(out), f.shape[0], f.shape[1], t.shape[0], _stream()), "eoe_clip_score")
    return out


class DsadLossFunction(torch.autograd.Function):
    """DSADTrainer.loss (dsad.py:17-21)"""

    @staticmethod
    def forward(ctx, feats, labels, nominal_label, inv_count):
        _chk(feats, labels)
        f = feats.contiguous().float()
        n, d = f.shape
        labels = labels.contiguous().to(torch.int64)
        loss = torch.empty(1, dtype=torch.float32, device=f.device)
        losses = torch.empty(n, dtype=torch.float32, device=f.device)
        inv = float(inv_count) if inv_count is not None else 1.0 / n
        check(lib.eoe_dsad_fwd(_p(f), _p(labels), int(nominal_label), _p(loss), _p(losses), n, d, inv, _stream()), "eoe_dsad_fwd")
        ctx.save_for_backward(f, labels)
        ctx.cfg = (int(nominal_label), inv)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gout):
        f, labels = ctx.saved_tensors
        nominal, inv = ctx.cfg
        df = torch.empty_like(f)
        gs = gout.contiguous().float().reshape(1)
        check(lib.eoe_dsad_bwd(_p(f), _p(labels), nominal, _p(gs), _p(df), f.shape[0], f.shape[1], inv * _grad_scale, _stream()), "eoe_dsad_bwd")
        return df, None, None, None


def dsad_loss(feats, labels, nominal_label=0, inv_count=None):
    return DsadLossFunction.apply(feats, labels, nominal_label, inv_count)


class DsvddLossFunction(torch.autograd.Function):
    """DSVDDTrainer.loss (dsvdd.py:26-27): mean squared distance to the fixed centre"""

    @staticmethod
    def forward(ctx, feats, center, inv_count):
        _chk(feats, center)
        f = feats.contiguous().float()
        n, d = f.shape
        c = center.detach().contiguous().float().reshape(-1)
        assert c.shape[0] == d
        loss = torch.empty(1, dtype=torch.float32, device=f.device)
        dists = torch.empty(n, dtype=torch.float32, device=f.device)
        inv = float(inv_count) if inv_count is not None else 1.0 / n
        check(lib.eoe_dsvdd_fwd(_p(f), _p(c), _p(loss), _p(dists), n, d, inv, _stream()), "eoe_dsvdd_fwd")
        ctx.save_for_backward(f, c)
        ctx.inv = inv
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gout):
        f, c = ctx.saved_tensors
        df = torch.empty_like(f)
        gs = gout.contiguous().float().reshape(1)
        check(lib.eoe_dsvdd_bwd(_p(f), _p(c), _p(gs), _p(df), f.shape[0], f.shape[1], ctx.inv * _grad_scale, _stream()), "eoe_dsvdd_bwd")
        return df, None, None


def dsvdd_loss(feats, center, inv_count=None):
    return DsvddLossFunction.apply(feats, center, inv_count)


def dsvdd_score(feats, center):
    """DSVDDTrainer.compute_anomaly_score (dsvdd.py:24-25)"""
    _chk(feats, center)
    f = feats.detach().contiguous().float()
    c = center.detach().contiguous().float().reshape(-1)
    out = torch.empty(f.shape[0], dtype=torch.float32, device=f.device)
    check(lib.eoe_dsvdd_fwd(_p(f), _p(c), None, _p(out), f.shape[0], f.shape[1], 1.0, _stream()), "eoe_dsvdd_fwd")
    return out


class FocalLossFunction(torch.autograd.Function):
    """FocalTrainer.loss (focal.py:11-24,34-36), gamma = 2, eps = 1e-7"""

    @staticmethod
    def forward(ctx, feats, labels, inv_count, gamma, eps):
        _chk(feats, labels)
        x = feats.contiguous().float().reshape(-1)
        n = x.shape[0]
        labels = labels.contiguous().to(torch.int64)
        loss = torch.empty(1, dtype=torch.float32, device=x.device)
        losses = torch.empty(n, dtype=torch.float32, device=x.device)
        inv = float(inv_count) if inv_count is not None else 1.0 / n
        check(lib.eoe_focal_fwd(_p(x), _p(labels), 0, _p(loss), None, _p(losses), n, inv, float(gamma), float(eps), _stream()),
              "eoe_focal_fwd")
        ctx.save_for_backward(x, labels)
        ctx.cfg = (inv, float(gamma), float(eps), feats.shape)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gout):
        x, labels = ctx.saved_tensors
        inv, gamma, eps, shape = ctx.cfg
        dx = torch.empty_like(x)
        gs = gout.contiguous().float().reshape(1)
        check(lib.eoe_focal_bwd(_p(x), _p(labels), _p(gs), _p(dx), x.shape[0], inv * _grad_scale, gamma, eps, _stream()), "eoe_focal_bwd")
        return dx.reshape(shape), None, None, None, None


def focal_loss(feats, labels, inv_count=None, gamma=2.0, eps=1e-7):
    return FocalLossFunction.apply(feats, labels, inv_count, gamma, eps)


# ------------------------------------------------------------------------------------------------ autograd: CNN backbone
BN_SCRATCH = (1024 + 3) * 2         # EOE_BN_SCRATCH(C) / C of include/eoe_hip.h: per-workgroup partial sums + totals (+ the sync-BN doubles)


def _conv_kp(cin: int, taps: int = 25) -> int:
    return (taps * cin + 63) // 64 * 64


def _conv_weight_copies(w: torch.Tensor, cpad=None):
    """16-bit copies of a conv weight, cached like `shadow`: [cout, Kp] in patch-column order, its transpose [Kp, cout]
    (materialised dgrad) and [cin, (taps reversed) x cout] (implicit stride-1 dgrad).  cpad: channels per tap in the column
    order when the gathered tensor is channel-padded (NHWC8 image of a 3-channel first layer)"""
    cpad = int(cpad or w.shape[1])
    key = ("conv", id(w), cpad)
    tag = (w._version, w.data_ptr(), _compute_dtype)
    capturing = torch.cuda.is_current_stream_capturing()           # see _Shadow.get
    hit = shadow.cache.get(key)
    if not capturing and hit is not None and hit[0]() is w and hit[1] == tag:
        return hit[2], hit[3], hit[4]
    if capturing and key in _capture_packs:                        # packed by this capture's batched refresh
        return _capture_packs[key]
    cout, cin, kh, kw = w.shape
    kp = _conv_kp(cpad, kh * kw)
    w16 = torch.empty((cout, kp), dtype=_compute_dtype, device=w.device)
    w16t = torch.empty((kp, cout), dtype=_compute_dtype, device=w.device)
    w16d = torch.empty((cin, kh * kw * cout), dtype=_compute_dtype, device=w.device)
    check(lib.eoe_conv_pack_weight(_p(w.detach().contiguous()), _p(w16), _p(w16t), _p(w16d), cout, cin, cpad, kh, kw, kp,
                                   dtype_code(_compute_dtype), _stream()), "eoe_conv_pack_weight")
    if not capturing:
        shadow.cache[key] = (weakref.ref(w, lambda _r, k=key, c=shadow.cache: c.pop(k, None)), tag, w16, w16t, w16d)
    return w16, w16t, w16d


_capture_packs = {}


def refresh_conv_weight_copies(weights):
    """bring the 16-bit copies of all conv weights in `weights` (tensors, or (tensor, cpad) pairs) up to date with ONE batched
    launch (`eoe_conv_pack_weight_multi`); the per-layer `_conv_weight_copies` calls of the step then hit the cache.  Under graph
    capture the batched pack becomes part of the graph and its outputs are handed to the per-layer calls of the same capture."""
    capturing = torch.cuda.is_current_stream_capturing()
    _capture_packs.clear()
    jobs, entries = [], []
    for item in weights:
        w, cpad = item if isinstance(item, tuple) else (item, None)
        cpad = int(cpad or w.shape[1])
        key = ("conv", id(w), cpad)
        tag = (w._version, w.data_ptr(), _compute_dtype)
        hit = shadow.cache.get(key)
        if not capturing and hit is not None and hit[0]() is w and hit[1] == tag:
            continue
        if not (w.is_cuda and w.is_contiguous() and w.dim() == 4):
            continue
        cout, cin, kh, kw = w.shape
        kp = _conv_kp(cpad, kh * kw)
        w16 = torch.empty((cout, kp), dtype=_compute_dtype, device=w.device)
        w16t = torch.empty((kp, cout), dtype=_compute_dtype, device=w.device)
        w16d = torch.empty((cin, kh * kw * cout), dtype=_compute_dtype, device=w.device)
        jobs.append(_lib.ConvPackJob(_p(w.detach()), _p(w16), _p(w16t), _p(w16d), cout, cin, cpad, kh, kw, kp))
        entries.append((key, w, tag, w16, w16t, w16d))
    if not jobs:
        return
    arr = (_lib.ConvPackJob * len(jobs))(*jobs)
    check(lib.eoe_conv_pack_weight_multi(arr, len(jobs), dtype_code(_compute_dtype), _stream()), "eoe_conv_pack_weight_multi")
    for key, w, tag, w16, w16t, w16d in entries:
        if capturing:                   # part of the graph: every replay re-packs; the per-layer calls of THIS capture pick these up
            _capture_packs[key] = (w16, w16t, w16d)
        else:
            shadow.cache[key] = (weakref.ref(w, lambda _r, k=key, c=shadow.cache: c.pop(k, None)), tag, w16, w16t, w16d)


def _stem_weight_copy(w: torch.Tensor):
    """16-bit [cout, ceil(kh/2)*64] copy of a 3-channel first-layer weight in the packed (ky, kx, c4) column order"""
    key = ("stem", id(w))
    tag = (w._version, w.data_ptr(), _compute_dtype)
    capturing = torch.cuda.is_current_stream_capturing()           # see _Shadow.get
    hit = shadow.cache.get(key)
    if not capturing and hit is not None and hit[0]() is w and hit[1] == tag:
        return hit[2]
    cout, _, kh, kw = w.shape
    w16 = torch.empty((cout, (kh + 1) // 2 * 64), dtype=_compute_dtype, device=w.device)
    check(lib.eoe_stem_pack_weight(_p(w.detach().contiguous()), _p(w16), cout, kh, kw, dtype_code(_compute_dtype), _stream()),
          "eoe_stem_pack_weight")
    if not capturing:
        shadow.cache[key] = (weakref.ref(w, lambda _r, k=key, c=shadow.cache: c.pop(k, None)), tag, w16)
    return w16


_implicit_conv = True


_fused_bn_stats = True


def set_fused_bn_stats(on: bool):
    """BatchNorm batch statistics from the conv GEMM's epilogue (default) or from a separate pass over its output"""
    global _fused_bn_stats
    _fused_bn_stats = bool(on)


def set_implicit_conv(on: bool):
    """A/B switch: True (default) = convolutions with cin % 64 == 0 fetch their patches inside the GEMM's LDS stage
    (implicit GEMM); False = every convolution materialises its patch matrix with eoe_im2col"""
    global _implicit_conv
    _implicit_conv = bool(on)


def _single_pixel(geo) -> bool:
    """a 1x1 input map convolved at stride 1 to a 1x1 output with the pixel inside the kernel window"""
    n, H, W, Cc, kh, kw, stride, pad, Ho, Wo = geo
    return H == 1 and W == 1 and Ho == 1 and Wo == 1 and stride == 1 and 0 <= pad < min(kh, kw) and Cc % 64 == 0


def conv_gemm_fwd(x16, w16, y, geo, bias=None, mode=1, colstats_ws=None, accumulate=False):
    """y[n*Ho*Wo, cout] (fp32) = patches(x16) @ w16^T without materialising the patches; x16 16-bit NHWC [n,H,W,C]
    (mode 2: the zero-padded NHWC4 image of eoe_stem_pack_image, packed k axis)"""
    n, H, W, Cc, kh, kw, stride, pad, Ho, Wo = geo
    M, K, N = n * Ho * Wo, (_conv_kp(Cc, kh * kw) if mode == 1 else (kh + 1) // 2 * 64), w16.shape[0]
    assert x16.is_contiguous() and w16.shape[1] == K and w16.stride(1) == 1 and y.shape == (M, N) and y.stride(1) == 1
    if mode == 1 and _single_pixel(geo):
        # a 1x1 map under a padded kernel (WideResNet's last stage on 32x32 inputs): only the tap that lands on the pixel sees data,
        # every other tap multiplies padding -- the same sum as a plain GEMM over that tap's column slice of w16, 1 / (kh*kw) of the
        # k-tiles (K = 4608 -> 512: 75 -> 10 us for 256 images)
        tap = pad * kw + pad
        return gemm_nt(x16.view(n, Cc), w16[:, tap * Cc:(tap + 1) * Cc], y, bias=bias, accumulate=accumulate, colstats_ws=colstats_ws)
    g = GemmArgs(_p(x16), _p(w16), _p(y), _p(bias), None, None, None, M, N, K, 0, w16.stride(0), y.stride(0), 0,
                 dtype_code(x16.dtype), EPI_NONE, 1 if y.dtype == torch.float32 else 0, 1 if accumulate else 0, 1.0, None, 0, mode,
                 _lib.ConvGeometry(*geo))
    if colstats_ws is not None:       # BatchNorm batch statistics from the epilogue: per-64-row partial (sum, sum of squares)
        g.workspace, g.workspace_bytes, g.colstats = _p(colstats_ws), colstats_ws.numel() * 4, 1
    check(lib.eoe_gemm_nt(C.byref(g), _stream()), "eoe_gemm_nt")
    return y


def conv_gemm_wgrad(x16, dy16, gT, geo, mode=1, dw=None):
    """gT[kh*kw*C, cout] (fp32) = patches(x16)^T @ dy16 without materialising the patches.  With `dw` (the weight gradient
    [cout, C, kh, kw] itself, mode 1) the launch's reduce kernel writes that layout directly and gT is not needed (`unpack_dw`)"""
    n, H, W, Cc, kh, kw, stride, pad, Ho, Wo = geo
    T, M, N = n * Ho * Wo, (kh * kw * Cc if mode == 1 else (kh + 1) // 2 * 64), dy16.shape[1]
    if dw is not None and mode == 1 and not _single_pixel(geo):
        assert x16.is_contiguous() and dy16.shape[0] == T and dy16.stride(1) == 1 and dw.shape == (N, Cc, kh, kw) and dw.is_contiguous()
        ws = scratch("tn_ws", (TN_WORKSPACE_BYTES,), torch.uint8, x16.device)
        g = GemmArgs(_p(x16), _p(dy16), _p(dw), None, None, None, None, M, N, T, 0, dy16.stride(0), N, 0, dtype_code(x16.dtype),
                     EPI_NONE, 1, 0, 1.0, _p(ws), TN_WORKSPACE_BYTES, mode, _lib.ConvGeometry(*geo))
        g.unpack_dw = 1
        check(lib.eoe_gemm_tn(C.byref(g), _stream()), "eoe_gemm_tn")
        return dw
    assert x16.is_contiguous() and dy16.shape[0] == T and dy16.stride(1) == 1 and gT.shape == (M, N) and gT.is_contiguous()
    if mode == 1 and _single_pixel(geo):
        # (see conv_gemm_fwd) the taps that only ever see padding have a zero gradient; the one on the pixel is a plain x^T dy
        tap = pad * kw + pad
        gT.zero_()
        return gemm_tn(x16.view(n, Cc), dy16, gT[tap * Cc:(tap + 1) * Cc])
    ws = scratch("tn_ws", (TN_WORKSPACE_BYTES,), torch.uint8, x16.device)
    g = GemmArgs(_p(x16), _p(dy16), _p(gT), None, None, None, None, M, N, T, 0, dy16.stride(0), N, 0, dtype_code(x16.dtype),
                 EPI_NONE, 1, 0, 1.0, _p(ws), TN_WORKSPACE_BYTES, mode, _lib.ConvGeometry(*geo))
    check(lib.eoe_gemm_tn(C.byref(g), _stream()), "eoe_gemm_tn")
    return gT


# ------------------------------------------------------------------------------------------------ parity mode (fp32 conv / linear)
_parity = False


def set_parity_mode(on: bool):
    """Parity mode: the convolutions and linear layers of the BatchNorm encoders (CNN32 / CNN28 / WideResNet, and
    `CustomNet.final_linear`) run as plain fp32 implicit GEMMs (csrc/parity.hip: fp32 activations and fp32 master weights as
    operands, no 16-bit rounding) instead of the 16-bit MFMA path.  A correctness instrument for the K-step trajectory tests
    (SURVEY.md section 8d "Parity run"), orders of magnitude slower than the fast path.  The ViT blocks are not affected: their
    fast path meets the 1e-3 bar at the benchmark batch (tests/test_gpu_parity_big.py)."""
    global _parity
    _parity = bool(on)


def parity_mode() -> bool:
    return _parity


# Measurement switch (round 5; tools/split_operand_noise.sh): EOE_PARITY_EMULATE_BITS=b rounds both operands of every parity-mode FORWARD
# convolution to b explicit mantissa bits before the exact fp32 product -- the arithmetic of a split-operand 16-bit MFMA product that keeps
# every cross term (b = 15: bf16 hi + lo; b = 21: fp16 hi + lo), i.e. a LOWER bound on the noise of the three-product form
# hi.hi + hi.lo + lo.hi.  Not a product mode: it only answers whether such a mode could hold the K_NOISE_PARITY bar.
PARITY_EMULATE_BITS = int(os.environ.get("EOE_PARITY_EMULATE_BITS", "0"))


def _round_mantissa(t: torch.Tensor, bits: int) -> torch.Tensor:
    drop = 23 - bits
    i = t.contiguous().view(torch.int32)
    return ((i + (1 << (drop - 1))) & ~((1 << drop) - 1)).view(torch.float32)


def _geo(n, H, W, C, kh, kw, stride, pad, Ho, Wo):
    from ._lib import ConvGeometry
    return ConvGeometry(n, H, W, C, kh, kw, stride, pad, Ho, Wo)


def _f32_colsum(x2d: torch.Tensor, out: torch.Tensor):
    """out[c] = sum_r x[r, c] of an fp32 matrix, atomics-free (eoe_colsum_f32; widths that are not multiples of 4 through eoe_cast_colsum)"""
    rows, cols = x2d.shape
    if cols % 4 == 0 and cols <= 4096:
        red = scratch("bn_red", (BN_SCRATCH * cols,), torch.float32, x2d.device)
        check(lib.eoe_colsum_f32(_p(x2d), _p(out), _p(red), rows, cols, 0, _stream()), "eoe_colsum_f32")
        return
    dummy = scratch("parity_colsum_dst", (rows * cols,), torch.float16, x2d.device)
    part = scratch("parity_colsum_part", (256 * cols,), torch.float32, x2d.device)
    check(lib.eoe_cast_colsum(_p(x2d), _p(dummy), _p(out), _p(part), rows, cols, _lib.EOE_F16, 0, _stream()), "eoe_cast_colsum")


class ConvBnActPoolParityFunction(torch.autograd.Function):
    """ConvBnActPoolFunction with the convolution in fp32 (parity mode); BatchNorm / activation / pooling are the same fp32
    kernels as on the fast path.  Same cfg tuple; never emits 16-bit copies."""

    @staticmethod
    def forward(ctx, x, conv_w, conv_b, bn_w, bn_b, rm, rv, nbt, cfg):
        _chk(x, conv_w, conv_b, bn_w, bn_b, rm, rv)
        training, eps, momentum, pool, is_image, mean, std, flat_out = cfg[:8]
        kh, kw, stride, pad = cfg[8] if len(cfg) > 8 else (5, 5, 1, 2)
        slope = float(cfg[9]) if len(cfg) > 9 else 0.01
        passthrough = bool(cfg[11]) if len(cfg) > 11 else False
        x_in = x
        x = x.contiguous().float()
        cout, cin = conv_w.shape[0], conv_w.shape[1]
        if is_image:
            n, _, Hi, Wi = x.shape
        else:
            n, Hi, Wi, _ = x.shape
        H, W = (Hi + 2 * pad - kh) // stride + 1, (Wi + 2 * pad - kw) // stride + 1
        M, dev = n * H * W, x.device
        geo = _geo(n, Hi, Wi, cin, kh, kw, stride, pad, H, W)
        w = conv_w.contiguous()
        y = torch.empty((M, cout), dtype=torch.float32, device=dev)
        ws = _parity_splitk_ws(dev)
        packed = bool(is_image and cin == 3 and cout % 4 == 0)
        if packed:
            # the 3-channel image normalised once into an fp32 NHWC4 map (4th channel and 4th weight channel zero): the first layer's forward
            # and weight gradient fetch whole pixels as float4 instead of gathering + normalising each element kh * kw times
            x4 = torch.empty((n, Hi, Wi, 4), dtype=torch.float32, device=dev)
            check(lib.eoe_pack_image_nhwc4(_p(x), _p(mean), _p(std), _p(x4), n, Hi, Wi, _stream()), "eoe_pack_image_nhwc4")
            w4 = scratch("parity_w4", (cout, 4, kh, kw), torch.float32, dev)
            w4.zero_()
            w4[:, :3].copy_(w)
            xq = x4
            if PARITY_EMULATE_BITS:
                xq = _round_mantissa(x4, PARITY_EMULATE_BITS)
                w4.copy_(_round_mantissa(w4, PARITY_EMULATE_BITS))
            w4f = scratch("parity_w4f", (kh * kw * 4, cout), torch.float32, dev)
            check(lib.eoe_conv_f32_pack_weights(_p(w4), _p(w4f), None, cout, 4, kh, kw, _stream()), "eoe_conv_f32_pack_weights")
            check(lib.eoe_conv_f32_fwd(_p(xq), 0, None, None, _p(w4), _p(conv_b), _p(y), _geo(n, Hi, Wi, 4, kh, kw, stride, pad, H, W), cout,
                                       _p(ws), PARITY_SPLITK_BYTES, _p(w4f), _stream()), "eoe_conv_f32_fwd")
            x = x4
        else:
            wkf = _parity_packed(conv_w, "f") if (not is_image and cout % 4 == 0) else None
            xq, wq = x, w
            if PARITY_EMULATE_BITS and not is_image:
                xq, wq, wkf = _round_mantissa(x, PARITY_EMULATE_BITS), _round_mantissa(w, PARITY_EMULATE_BITS), None      # (the kernel then reads the unpacked weights)
            check(lib.eoe_conv_f32_fwd(_p(xq), 1 if is_image else 0, _p(mean) if is_image else None, _p(std) if is_image else None, _p(wq),
                                       _p(conv_b), _p(y), geo, cout, _p(ws), PARITY_SPLITK_BYTES, _p(wkf), _stream()), "eoe_conv_f32_fwd")
        stats = torch.empty(2 * cout, dtype=torch.float32, device=dev)
        sums = scratch("bn_sums", (BN_SCRATCH * cout,), torch.float32, dev)
        check(lib.eoe_bn_stats(_p(y), _p(sums), _p(stats), _p(rm), _p(rv), _p(nbt), M, cout, float(eps), float(momentum),
                               1 if training else 0, _stream()), "eoe_bn_stats")
        idx = None
        code = dtype_code(_compute_dtype)
        if isinstance(pool, tuple):
            pk, pstride, ppad = pool
            Ho, Wo = (H + 2 * ppad - pk) // pstride + 1, (W + 2 * ppad - pk) // pstride + 1
            out = torch.empty((n, Ho, Wo, cout), dtype=torch.float32, device=dev)
            idx = torch.empty((n, Ho, Wo, cout), dtype=torch.uint8, device=dev)
            check(lib.eoe_bn_act_maxpool_fwd(_p(y), _p(stats), _p(bn_w), _p(bn_b), _p(out), None, _p(idx), n, H, W, cout, pk,
                                             pstride, ppad, slope, code, _stream()), "eoe_bn_act_maxpool_fwd")
        else:
            Ho, Wo = H // pool, W // pool
            out = torch.empty((n, cout * Ho * Wo) if flat_out else (n, Ho, Wo, cout), dtype=torch.float32, device=dev)
            check(lib.eoe_bn_act_pool_fwd(_p(y), _p(stats), _p(bn_w), _p(bn_b), _p(out), None, n, H, W, cout, pool,
                                          1 if flat_out else 0, 1, slope, code, _stream()), "eoe_bn_act_pool_fwd")
        ctx.save_for_backward(x, y, stats, conv_w, conv_b, bn_w, bn_b, idx, mean if is_image else None, std if is_image else None)
        ctx.cfg = (n, H, W, cin, cout, pool, flat_out, training, is_image, Hi, Wi, kh, kw, stride, pad, slope)
        ctx.packed = packed
        ctx.passthrough = passthrough
        if not passthrough:
            return out
        ctx.set_materialize_grads(False)
        return out, x_in

    @staticmethod
    def backward(ctx, dout, *more):
        d_pass = more[-1] if (ctx.passthrough and more) else None
        if dout is None:
            return (d_pass,) + (None,) * 8
        x, y, stats, conv_w, conv_b, bn_w, bn_b, idx, mean, std = ctx.saved_tensors
        n, H, W, cin, cout, pool, flat_out, training, is_image, Hi, Wi, kh, kw, stride, pad, slope = ctx.cfg
        packed = getattr(ctx, "packed", False)
        dev = y.device
        M = n * H * W
        dout = dout.contiguous().float()
        dy = torch.empty((M, cout), dtype=torch.float32, device=dev)
        dg = _grad_target(bn_w) if bn_w is not None else None
        db = _grad_target(bn_b) if bn_b is not None else None
        red = scratch("bn_red", (BN_SCRATCH * cout,), torch.float32, dev)
        if isinstance(pool, tuple):
            check(lib.eoe_bn_act_maxpool_bwd(_p(y), _p(stats), _p(bn_w), _p(bn_b), _p(dout), _p(idx), _p(red), _p(dy), _p(dg), _p(db),
                                             n, H, W, cout, pool[0], pool[1], pool[2], 1 if training else 0, slope, _lib.EOE_F32, _stream()),
                  "eoe_bn_act_maxpool_bwd")
        else:
            check(lib.eoe_bn_act_pool_bwd(_p(y), _p(stats), _p(bn_w), _p(bn_b), _p(dout), _p(red), _p(dy), 1, _p(dg), _p(db), n, H,
                                          W, cout, pool, 1 if flat_out else 0, 1 if training else 0, 0, slope,
                                          dtype_code(_compute_dtype), _stream()), "eoe_bn_act_pool_bwd")
        geo = _geo(n, Hi, Wi, cin, kh, kw, stride, pad, H, W)
        w = conv_w.contiguous()
        dw = _grad_target(conv_w)
        _side_ctx = _conv_wgrad_side_begin(dy, x) if CONV_ASYNC_WGRAD else None      # (see _conv_wgrad_side_begin: under dgrad + the next layer)
        if packed:                                   # x = the NHWC4 image saved by forward: gradient of the zero-padded [cout, 4, kh, kw] weight
            geo4 = _geo(n, Hi, Wi, 4, kh, kw, stride, pad, H, W)
            ws_bytes = int(lib.eoe_conv_f32_wgrad_workspace(geo4, cout))
            ws = scratch("parity_wgrad_ws", (ws_bytes // 4,), torch.float32, dev)
            dw4 = scratch("parity_dw4", (cout, 4, kh, kw), torch.float32, dev)
            check(lib.eoe_conv_f32_wgrad(_p(x), 0, None, None, _p(dy), _p(dw4), geo4, cout, _p(ws), ws_bytes, _stream()), "eoe_conv_f32_wgrad")
            dw.copy_(dw4[:, :3])
        else:
            ws_bytes = int(lib.eoe_conv_f32_wgrad_workspace(geo, cout))
            ws = scratch("parity_wgrad_ws", (ws_bytes // 4,), torch.float32, dev)
            check(lib.eoe_conv_f32_wgrad(_p(x), 1 if is_image else 0, _p(mean), _p(std), _p(dy), _p(dw), geo, cout, _p(ws), ws_bytes,
                                         _stream()), "eoe_conv_f32_wgrad")
        if _side_ctx is not None:
            _conv_wgrad_side_end(_side_ctx)
        dcb = None
        if conv_b is not None:
            dcb = _grad_target(conv_b)
            _f32_colsum(dy, dcb)
        dx = None
        if ctx.needs_input_grad[0] and not is_image:
            acc = (d_pass is not None and d_pass.dtype == torch.float32 and d_pass.is_contiguous()
                   and d_pass.shape == (n, Hi, Wi, cin))
            dx = d_pass if acc else torch.empty((n, Hi, Wi, cin), dtype=torch.float32, device=dev)
            wkd = _parity_packed(conv_w, "d") if cin % 4 == 0 else None
            check(lib.eoe_conv_f32_dgrad(_p(dy), _p(w), _p(dx), geo, cout, 1 if acc else 0, _p(_parity_splitk_ws(dev)), PARITY_SPLITK_BYTES,
                                         _p(wkd), _stream()), "eoe_conv_f32_dgrad")
            if acc:
                d_pass = None
        if d_pass is not None:
            dx = d_pass if dx is None else dx.add_(d_pass)
        return dx, dw, dcb, dg, db, None, None, None, None


PARITY_SPLITK_BYTES = 16 << 20
_parity_packs = {}            # id(weight) -> (weakref, version, wf, wd): k-major fp32 copies of a convolution / linear weight, per version


def _parity_packed(w, want, shape4=None):
    """the k-major fp32 copy (`want` = "f": forward [kh*kw*cin, cout]; "d": dgrad [kh*kw*cout, cin]) of a weight [cout, cin, kh, kw]
    (`shape4`: how to read a Linear weight [out, in] as one) for the exact-fp32 kernels (eoe_conv_f32_pack_weights): made once per
    weight version (the optimiser bumps it), both layouts in one launch"""
    key = id(w)
    hit = _parity_packs.get(key)
    if hit is not None and hit[0]() is w and hit[1] == (w._version, w.data_ptr()) and hit[2].device == w.device:
        return hit[2] if want == "f" else hit[3]
    cout, cin, kh, kw = shape4 if shape4 is not None else w.shape
    wc = w.detach().contiguous()
    if hit is not None and hit[0]() is w and hit[2].numel() == wc.numel() and hit[2].device == w.device:
        wf, wd = hit[2], hit[3]                                       # reuse the buffers of the previous version
    else:
        wf = torch.empty((kh * kw * cin, cout), dtype=torch.float32, device=w.device)
        wd = torch.empty((kh * kw * cout, cin), dtype=torch.float32, device=w.device)
    check(lib.eoe_conv_f32_pack_weights(_p(wc), _p(wf), _p(wd), cout, cin, kh, kw, _stream()), "eoe_conv_f32_pack_weights")
    if not torch.cuda.is_current_stream_capturing():
        _parity_packs[key] = (weakref.ref(w), (w._version, w.data_ptr()), wf, wd)
    return wf if want == "f" else wd


def _parity_splitk_ws(dev):
    """slab partials of the fp32 forward / dgrad kernels when they cut a long reduction over few output tiles (eoe_conv_f32_fwd)"""
    return scratch("parity_splitk_ws", (PARITY_SPLITK_BYTES // 4,), torch.float32, dev)


class LinearParityFunction(torch.autograd.Function):
    """y = x @ W^T + b in fp32 (parity mode): a 1x1 convolution on 1x1 images through the fp32 kernels"""

    @staticmethod
    def forward(ctx, x, weight, bias):
        _chk(x, weight, bias)
        x2 = x.reshape(-1, x.shape[-1]).contiguous().float()
        M, K = x2.shape
        N = weight.shape[0]
        y = torch.empty((M, N), dtype=torch.float32, device=x.device)
        wkf = _parity_packed(weight, "f", (N, K, 1, 1)) if N % 4 == 0 else None      # W^T, once per weight version
        check(lib.eoe_conv_f32_fwd(_p(x2), 0, None, None, _p(weight.contiguous()), _p(bias), _p(y), _geo(M, 1, 1, K, 1, 1, 1, 0, 1, 1), N,
                                   _p(_parity_splitk_ws(x.device)), PARITY_SPLITK_BYTES, _p(wkf), _stream()), "eoe_conv_f32_fwd")
        ctx.save_for_backward(x2, weight, bias)
        ctx.in_shape, ctx.x_dtype = x.shape, x.dtype
        return y.reshape(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, weight, bias = ctx.saved_tensors
        M, K = x2.shape
        N = weight.shape[0]
        dy2 = dy.reshape(M, N).contiguous().float()
        geo = _geo(M, 1, 1, K, 1, 1, 1, 0, 1, 1)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((M, K), dtype=torch.float32, device=dy.device)
            wc = weight.contiguous()             # a 1 x 1 kernel's dgrad layout [cout][cin] is the weight itself
            check(lib.eoe_conv_f32_dgrad(_p(dy2), _p(wc), _p(dx), geo, N, 0, _p(_parity_splitk_ws(dy.device)), PARITY_SPLITK_BYTES,
                                         _p(wc) if K % 4 == 0 else None, _stream()), "eoe_conv_f32_dgrad")
            dx = dx.reshape(ctx.in_shape).to(ctx.x_dtype)
        if ctx.needs_input_grad[1]:
            dw = _grad_target(weight)
            ws_bytes = int(lib.eoe_conv_f32_wgrad_workspace(geo, N))
            ws = scratch("parity_wgrad_ws", (ws_bytes // 4,), torch.float32, dy.device)
            check(lib.eoe_conv_f32_wgrad(_p(x2), 0, None, None, _p(dy2), _p(dw), geo, N, _p(ws), ws_bytes, _stream()), "eoe_conv_f32_wgrad")
        if bias is not None and ctx.needs_input_grad[2]:
            db = _grad_target(bias)
            _f32_colsum(dy2, db)
        return dx, dw, db


# EOE_CONV_Y16=1 (bench.py --conv-y16): the convolution output between the GEMM and BatchNorm in fp16 instead of fp32
# (ConvBnActPoolFunction.forward).  A speed option of the 16-bit fast mode, off by default: WideResNet-224 12.3 -> 11.45 ms per step, at the
# price of one more rounding point in front of every BatchNorm -- on the full-batch fixture the loss leaves the 1e-3 the default fast
# mode holds (3.2e-3 at worst, reference noise 2.7e-4); AUC stays within 2.4e-4.
CONV_Y16 = os.environ.get("EOE_CONV_Y16", "0") != "0"


class ConvBnActPoolFunction(torch.autograd.Function):
    """conv (+ bias) -> BatchNorm2d -> act -> MaxPool(pool), one layer per call: conv5x5(pad 2) + LeakyReLU(0.01) + pool 2
    for `cnn.py:73-82` (the default when cfg has 8 entries); cfg[8] = (kh, kw, stride, pad) and cfg[9] = the activation's
    negative slope (0 = ReLU, 1 = none) give the conv->bn(->relu) units of `resnet.py:93-95,133-141`; cfg[10] = also emit a
    16-bit copy of the output (returned second, non-differentiable) for a following convolution.
    Input: the fp32 NCHW image batch (first layer; optional fused Normalize) or an fp32 NHWC activation (a 16-bit copy
    attached to it as `._eoe16` by the producing op is used instead of casting); output: fp32 NHWC [n, Ho/p, Wo/p, cout], or
    the reference's NCHW-flattened [n, cout*(Ho/p)*(Wo/p)] (`cnn.py:83`) if flat_out.
    With cin % 64 == 0 the patch matrix is never built (implicit GEMM: forward, wgrad, and the dgrad of stride-1 convs);
    otherwise (the 3-channel first layers, CNN32's 32-channel layer) eoe_im2col materialises it once for forward + wgrad."""

    @staticmethod
    def forward(ctx, x, conv_w, conv_b, bn_w, bn_b, rm, rv, nbt, cfg):
        _chk(x, conv_w, conv_b, bn_w, bn_b, rm, rv)
        training, eps, momentum, pool, is_image, mean, std, flat_out = cfg[:8]
        kh, kw, stride, pad = cfg[8] if len(cfg) > 8 else (5, 5, 1, 2)
        slope = float(cfg[9]) if len(cfg) > 9 else 0.01
        # cfg[11]: also return the input itself (third output).  A residual block hands THAT to its junction as the shortcut,
        # so the junction's gradient for the shortcut arrives here (backward's d_pass) and the dgrad GEMM accumulates onto
        # it -- instead of autograd adding the two gradients of the block input with a separate elementwise pass
        passthrough = bool(cfg[11]) if len(cfg) > 11 else False
        x_in = x
        x16 = getattr(x, "_eoe16", None)
        x = x.contiguous().float()
        cout, cin = conv_w.shape[0], conv_w.shape[1]
        if is_image:
            n, _, Hi, Wi = x.shape
        else:
            n, Hi, Wi, _ = x.shape
        H, W = (Hi + 2 * pad - kh) // stride + 1, (Wi + 2 * pad - kw) // stride + 1       # conv output grid
        M, kp, dev, dt = n * H * W, _conv_kp(cin, kh * kw), x.device, _compute_dtype
        code = dtype_code(dt)
        implicit = _implicit_conv and (not is_image) and (cin % 64 == 0 or cin in (8, 16, 32))
        stem = _implicit_conv and is_image and cin == 3 and kw <= 8 and stride % 2 == 0
        img8 = _implicit_conv and is_image and cin == 3 and not stem           # NHWC8 image, per-piece tap decoding
        # training: the conv GEMM's epilogue leaves per-64-row (sum, sum of squares) of y in `part`, so that the batch
        # statistics need no pass over y (eoe_bn_stats_partials); eval: running statistics
        R = (M + 63) // 64
        part = scratch("bn_part", (R * 2 * cout,), torch.float32, dev) if (training and _fused_bn_stats and cout % 16 == 0) else None
        # CONV_Y16: with the statistics taken from the GEMM's fp32 accumulators, y itself is written (and saved for backward) in the 16-bit
        # compute type: the BatchNorm forward pass and the two backward passes read half the bytes, the GEMM's epilogue stores half
        # (fp16 only: 11 significant bits in front of a normalisation are enough, bfloat16's 8 are not -- a BatchNorm input of mean 3 sigma
        #  would carry 1e-2 sigma of rounding)
        y16mode = CONV_Y16 and part is not None and dt == torch.float16
        y = torch.empty((M, cout), dtype=dt if y16mode else torch.float32, device=dev)
        ycode = code | (_lib.EOE_Y16 if y16mode else 0)
        if stem:
            # packed first layer: normalised 16-bit NHWC4 image with physical zero padding, gathered inside the GEMM
            Hp, Wp = (H - 1) * stride + (kh + 1) // 2 * 2, ((W - 1) * stride + 8 + 1) // 2 * 2
            Hp, Wp = max(Hp, Hi + pad), max(Wp, Wi + pad + (Wi + pad) % 2)
            operand = torch.empty((n, Hp, Wp, 4), dtype=dt, device=dev)
            check(lib.eoe_stem_pack_image(_p(x), _p(mean), _p(std), _p(operand), n, Hi, Wi, Hp, Wp, pad, 4, code, _stream()),
                  "eoe_stem_pack_image")
            implicit = 2
            conv_gemm_fwd(operand, _stem_weight_copy(conv_w), y, (n, Hp, Wp, 4, kh, kw, stride, 0, H, W), bias=conv_b, mode=2,
                          colstats_ws=part)
        elif img8:
            operand = torch.empty((n, Hi, Wi, 8), dtype=dt, device=dev)
            check(lib.eoe_stem_pack_image(_p(x), _p(mean), _p(std), _p(operand), n, Hi, Wi, Hi, Wi, 0, 8, code, _stream()),
                  "eoe_stem_pack_image")
            implicit = 3
            w16, _, _ = _conv_weight_copies(conv_w, 8)
            conv_gemm_fwd(operand, w16, y, (n, Hi, Wi, 8, kh, kw, stride, pad, H, W), bias=conv_b, colstats_ws=part)
        elif implicit:
            w16, _, _ = _conv_weight_copies(conv_w)
            if x16 is None or x16.dtype != dt or x16.shape != x.shape:
                if getattr(x_in, "_eoe_unwritten", False):
                    raise RuntimeError("this activation only exists as its 16-bit copy (cfg[12] of the producing layer) and that copy does not fit")
                x16 = cast16(x.view(-1, cin)).view(n, Hi, Wi, cin)
            operand = x16
            conv_gemm_fwd(x16, w16, y, (n, Hi, Wi, cin, kh, kw, stride, pad, H, W), bias=conv_b, colstats_ws=part)
        else:
            w16, _, _ = _conv_weight_copies(conv_w)
            operand = torch.empty((M, kp), dtype=dt, device=dev)
            check(lib.eoe_im2col(_p(x), 1 if is_image else 2, _p(mean), _p(std), _p(operand), n, cin, Hi, Wi, kh, kw, stride, pad,
                                 kp, code, _stream()), "eoe_im2col")
            gemm_nt(operand, w16, y, bias=conv_b, colstats_ws=part)
        stats = torch.empty(2 * cout, dtype=torch.float32, device=dev)
        sums = scratch("bn_sums", (BN_SCRATCH * cout,), torch.float32, dev)
        if part is not None:
            check(lib.eoe_bn_stats_partials(_p(part), R, _p(sums), _p(stats), _p(rm), _p(rv), _p(nbt), M, cout, float(eps),
                                            float(momentum), _stream()), "eoe_bn_stats_partials")
        else:
            check(lib.eoe_bn_stats(_p(y), _p(sums), _p(stats), _p(rm), _p(rv), _p(nbt), M, cout, float(eps), float(momentum),
                                   1 if training else 0, _stream()), "eoe_bn_stats")
        want16 = bool(cfg[10]) if len(cfg) > 10 else False
        idx = None
        if isinstance(pool, tuple):
            # overlapping MaxPool2d(k, s, p) fused behind BN + act: the pre-pool activation is never written (resnet.py:93-96)
            pk, pstride, ppad = pool
            Ho, Wo = (H + 2 * ppad - pk) // pstride + 1, (W + 2 * ppad - pk) // pstride + 1
            out = torch.empty((n, Ho, Wo, cout), dtype=torch.float32, device=dev)
            out16 = torch.empty((n, Ho, Wo, cout), dtype=dt, device=dev) if (want16 and _implicit_conv) else None
            idx = torch.empty((n, Ho, Wo, cout), dtype=torch.uint8, device=dev)
            check(lib.eoe_bn_act_maxpool_fwd(_p(y), _p(stats), _p(bn_w), _p(bn_b), _p(out), _p(out16), _p(idx), n, H, W, cout, pk,
                                             pstride, ppad, slope, ycode, _stream()), "eoe_bn_act_maxpool_fwd")
        else:
            Ho, Wo = H // pool, W // pool
            out = torch.empty((n, cout * Ho * Wo) if flat_out else (n, Ho, Wo, cout), dtype=torch.float32, device=dev)
            # a 16-bit copy of the output for the next convolution's implicit GEMM (saves that layer a cast pass)
            out16 = torch.empty((n, Ho, Wo, cout), dtype=dt, device=dev) if (want16 and _implicit_conv and not flat_out) else None
            only16 = out16 is not None and len(cfg) > 12 and bool(cfg[12])
            if only16:
                # the output feeds ONE consumer that reads the 16-bit copy only (a BasicBlock's conv1 -> bn1 -> relu -> conv2): the fp32
                # tensor is allocated for autograd's bookkeeping but never written (4 of the pass's 10 bytes per element)
                check(lib.eoe_bn_act_pool_fwd(_p(y), _p(stats), _p(bn_w), _p(bn_b), _p(out16), None, n, H, W, cout, pool,
                                              0, 0, slope, ycode, _stream()), "eoe_bn_act_pool_fwd")
            else:
                check(lib.eoe_bn_act_pool_fwd(_p(y), _p(stats), _p(bn_w), _p(bn_b), _p(out), _p(out16), n, H, W, cout, pool,
                                              1 if flat_out else 0, 1, slope, ycode, _stream()), "eoe_bn_act_pool_fwd")
        ctx.save_for_backward(operand, y, stats, conv_w, conv_b, bn_w, bn_b, idx)
        ctx.cfg = (n, H, W, cin, cout, kp, pool, flat_out, training, is_image, Hi, Wi, kh, kw, stride, pad, slope, implicit)
        ctx.has16, ctx.passthrough = out16 is not None, passthrough
        if out16 is None and not passthrough:
            return out
        ctx.set_materialize_grads(False)       # else autograd zero-fills a gradient for the 16-bit copy every step
        outs = [out]
        if out16 is not None:
            ctx.mark_non_differentiable(out16)
            outs.append(out16)
        if passthrough:
            outs.append(x_in)
        return tuple(outs)

    @staticmethod
    def backward(ctx, dout, *more):
        d_pass = more[-1] if (ctx.passthrough and more) else None
        if dout is None:                         # only the pass-through output was used downstream
            return (d_pass,) + (None,) * 8
        operand, y, stats, conv_w, conv_b, bn_w, bn_b, idx = ctx.saved_tensors
        n, H, W, cin, cout, kp, pool, flat_out, training, is_image, Hi, Wi, kh, kw, stride, pad, slope, implicit = ctx.cfg
        dev, dt = y.device, operand.dtype
        code = dtype_code(dt)
        ycode = code | (_lib.EOE_Y16 if y.dtype != torch.float32 else 0)
        M = n * H * W
        dout = dout.contiguous().float()
        dy16 = torch.empty((M, cout), dtype=dt, device=dev)
        dg = _grad_target(bn_w) if bn_w is not None else None
        db = _grad_target(bn_b) if bn_b is not None else None
        red = scratch("bn_red", (BN_SCRATCH * cout,), torch.float32, dev)
        if isinstance(pool, tuple):
            check(lib.eoe_bn_act_maxpool_bwd(_p(y), _p(stats), _p(bn_w), _p(bn_b), _p(dout), _p(idx), _p(red), _p(dy16), _p(dg), _p(db),
                                             n, H, W, cout, pool[0], pool[1], pool[2], 1 if training else 0, slope, ycode, _stream()),
                  "eoe_bn_act_maxpool_bwd")
        else:
            check(lib.eoe_bn_act_pool_bwd(_p(y), _p(stats), _p(bn_w), _p(bn_b), _p(dout), _p(red), _p(dy16), 0, _p(dg), _p(db), n, H,
                                          W, cout, pool, 1 if flat_out else 0, 1 if training else 0, 0, slope, ycode, _stream()),
                  "eoe_bn_act_pool_bwd")
        dw = _grad_target(conv_w)
        _side_ctx = _conv_wgrad_side_begin(dy16, operand) if CONV_ASYNC_WGRAD else None
        if implicit == 2:
            _, Hp, Wp, _ = operand.shape
            gT = torch.empty(((kh + 1) // 2 * 64, cout), dtype=torch.float32, device=dev)
            conv_gemm_wgrad(operand, dy16, gT, (n, Hp, Wp, 4, kh, kw, stride, 0, H, W), mode=2)
            check(lib.eoe_stem_unpack_wgrad(_p(gT), _p(dw), cout, kh, kw, _stream()), "eoe_stem_unpack_wgrad")
        elif implicit:
            cg = 8 if implicit == 3 else cin          # channels per tap of the gathered tensor (NHWC8 image: 8)
            geo_w = (n, Hi, Wi, cg, kh, kw, stride, pad, H, W)
            if cg == cin and dw.is_contiguous() and not _single_pixel(geo_w):
                conv_gemm_wgrad(operand, dy16, None, geo_w, dw=dw)            # the reduce kernel writes dw's own layout
            else:
                gT = torch.empty((kh * kw * cg, cout), dtype=torch.float32, device=dev)
                conv_gemm_wgrad(operand, dy16, gT, geo_w)
                check(lib.eoe_conv_unpack_wgrad(_p(gT), _p(dw), cout, cin, cg, kh, kw, kh * kw * cg, 1, 0, _stream()),
                      "eoe_conv_unpack_wgrad")
        else:
            g = torch.empty((cout, kp), dtype=torch.float32, device=dev)
            gemm_tn(dy16, operand, g)
            check(lib.eoe_conv_unpack_wgrad(_p(g), _p(dw), cout, cin, cin, kh, kw, kp, 0, 0, _stream()), "eoe_conv_unpack_wgrad")
        if _side_ctx is not None:
            _conv_wgrad_side_end(_side_ctx)
        dcb = None
        if conv_b is not None:
            dcb = _grad_target(conv_b)
            colsum(dy16, dcb)
        dx = None
        if ctx.needs_input_grad[0] and not is_image:
            _, w16t, w16d = _conv_weight_copies(conv_w)
            acc = (d_pass is not None and d_pass.dtype == torch.float32 and d_pass.is_contiguous()
                   and d_pass.shape == (n, Hi, Wi, cin))
            if _implicit_conv and stride == 1 and kh == kw and cout % 64 == 0:
                # dx = conv of dy with the flipped kernel: the same implicit GEMM, gathering from dy16 [n, H, W, cout];
                # with a shortcut gradient it accumulates onto that tensor (C += in the epilogue)
                dx = d_pass if acc else torch.empty((n, Hi, Wi, cin), dtype=torch.float32, device=dev)
                conv_gemm_fwd(dy16.view(n, H, W, cout), w16d, dx.view(-1, cin),
                              (n, H, W, cout, kh, kw, 1, kh - 1 - pad, Hi, Wi), accumulate=acc)
                d_pass = None
            else:
                # materialised dgrad (the stride-2 convolutions): patches of dx, gathered back; onto the shortcut gradient if one came
                dx = d_pass if acc else torch.empty((n, Hi, Wi, cin), dtype=torch.float32, device=dev)
                dpatches = torch.empty((M, kp), dtype=dt, device=dev)
                gemm_nt(dy16, w16t, dpatches)
                check(lib.eoe_col2im(_p(dpatches), _p(dx), n, cin, Hi, Wi, kh, kw, stride, pad, kp, code, 1 if acc else 0, _stream()),
                      "eoe_col2im")
                if acc:
                    d_pass = None
        if d_pass is not None:
            dx = d_pass if dx is None else dx.add_(d_pass)
        return dx, dw, dcb, dg, db, None, None, None, None


# A convolution's weight-gradient GEMM (+ its unpack / reduce kernels) on a side stream: nothing in the backward sweep reads the weight
# gradients, so the launch runs under the layer's dgrad GEMM and the next layer's HBM-bound BatchNorm / CBAM backward instead of in front of
# them (WideResNet-224: 12.9 -> 12.67 ms per step interleaved, the same bits).  What it reads (dY, the saved operand) is kept alive until ONE
# join at the end of the backward pass (an autograd-engine callback).  Off while a stream is being captured and while a gradient arena's
# bucket hooks are installed (they send a gradient the moment autograd has it): `async_wgrad_blockers`.  EOE_CONV_ASYNC_WGRAD=0: A/B.
CONV_ASYNC_WGRAD = os.environ.get("EOE_CONV_ASYNC_WGRAD", "1") != "0"
async_wgrad_blockers = 0
_conv_side = None
_conv_keep = []


def _conv_wgrad_side_begin(*tensors):
    global _conv_side
    if async_wgrad_blockers > 0 or torch.cuda.is_current_stream_capturing():
        return None
    if _conv_side is None:
        _conv_side = torch.cuda.Stream()
    main = torch.cuda.current_stream()
    ev = torch.cuda.Event()
    ev.record(main)
    _conv_side.wait_event(ev)
    _conv_keep.extend(tensors)                  # what the launch reads stays alive until the join
    cm = torch.cuda.stream(_conv_side)
    cm.__enter__()
    torch.autograd.Variable._execution_engine.queue_callback(_conv_wgrad_join)      # every layer queues it (idempotent, see vit_side_join)
    return cm


def _conv_wgrad_side_end(cm):
    cm.__exit__(None, None, None)


def _conv_wgrad_join():
    if _conv_side is not None and _conv_keep:
        torch.cuda.current_stream().wait_stream(_conv_side)
    _conv_keep.clear()


def conv_bn_act_pool(x, conv_w, conv_b, bn_w, bn_b, rm, rv, nbt, cfg):
    """ConvBnActPoolFunction + the 16-bit copy of its output attached as `._eoe16` (consumed by the next convolution)"""
    if _parity:
        return ConvBnActPoolParityFunction.apply(x, conv_w, conv_b, bn_w, bn_b, rm, rv, nbt, cfg)
    r = ConvBnActPoolFunction.apply(x, conv_w, conv_b, bn_w, bn_b, rm, rv, nbt, cfg)
    passthrough = bool(cfg[11]) if len(cfg) > 11 else False
    if not isinstance(r, tuple):
        return r
    out = r[0]
    if len(r) - (1 if passthrough else 0) == 2:
        out._eoe16 = r[1]
        if len(cfg) > 12 and cfg[12] and not (isinstance(cfg[3], tuple) or cfg[7]):
            out._eoe_unwritten = True          # (ConvBnActPoolFunction.forward, only16: the values live in the 16-bit copy alone)
    if passthrough:
        x16 = getattr(x, "_eoe16", None)
        if x16 is not None:
            r[-1]._eoe16 = x16          # the handed-through input keeps its 16-bit copy (a down-sampling conv reads it next)
        return out, r[-1]
    return out


class BnActFunction(torch.autograd.Function):
    """BatchNorm1d -> LeakyReLU(0.01) on an fp32 [n, C] matrix (`cnn.py:84-85`)"""

    @staticmethod
    def forward(ctx, y, bn_w, bn_b, rm, rv, nbt, cfg):
        _chk(y, bn_w, bn_b, rm, rv)
        training, eps, momentum = cfg
        y = y.contiguous().float()
        n, C = y.shape
        dev = y.device
        stats = torch.empty(2 * C, dtype=torch.float32, device=dev)
        sums = scratch("bn_sums", (BN_SCRATCH * C,), torch.float32, dev)
        check(lib.eoe_bn_stats(_p(y), _p(sums), _p(stats), _p(rm), _p(rv), _p(nbt), n, C, float(eps), float(momentum),
                               1 if training else 0, _stream()), "eoe_bn_stats")
        out = torch.empty_like(y)
        check(lib.eoe_bn_act_pool_fwd(_p(y), _p(stats), _p(bn_w), _p(bn_b), _p(out), None, n, 1, 1, C, 1, 0, 1, 0.01,
                                      dtype_code(_compute_dtype), _stream()), "eoe_bn_act_pool_fwd")
        ctx.save_for_backward(y, stats, bn_w, bn_b)
        ctx.training = training
        return out

    @staticmethod
    def backward(ctx, dout):
        y, stats, bn_w, bn_b = ctx.saved_tensors
        n, C = y.shape
        dev = y.device
        dout = dout.contiguous().float()
        dy = torch.empty_like(y)
        dg = _grad_target(bn_w) if bn_w is not None else None
        db = _grad_target(bn_b) if bn_b is not None else None
        red = scratch("bn_red", (BN_SCRATCH * C,), torch.float32, dev)
        check(lib.eoe_bn_act_pool_bwd(_p(y), _p(stats), _p(bn_w), _p(bn_b), _p(dout), _p(red), _p(dy), 1, _p(dg), _p(db), n, 1, 1,
                                      C, 1, 0, 1 if ctx.training else 0, 0, 0.01, dtype_code(_compute_dtype), _stream()),
              "eoe_bn_act_pool_bwd")
        return dy, dg, db, None, None, None, None
