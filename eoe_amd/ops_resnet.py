"""autograd Functions for the WideResNet + CBAM encoder (reference `src/eoe/models/resnet.py:85-149`,
`src/eoe/models/cbam.py:31-107`): every forward / backward is one or a few C calls into libeoe_hip.so
(`include/eoe_hip.h`, section "WideResNet + CBAM").  Activations are fp32 NHWC between the ops; convolutions go
through `ops.ConvBnActPoolFunction` (im2col + MFMA GEMM + fused BatchNorm/activation)."""
import ctypes as C

import torch

from . import _lib
from ._lib import check, lib
from . import ops
SGATE_RED = 2 + 2 * 512         # EOE_SGATE_RED (include/eoe_hip.h)
from .ops import BN_SCRATCH, _chk, _grad_target, _p, _stream, scratch, dtype_code


import os as _os

FUSE_CBAM = _os.environ.get("EOE_FUSE_CBAM", "1") != "0"        # BasicBlock: ChannelGate + SpatialGate + residual junction through CbamJunctionFunction (0: the two units of round 2, A/B)


class MaxPoolFunction(torch.autograd.Function):
    """nn.MaxPool2d(k, stride, pad) on fp32 NHWC (`resnet.py:35,96`)"""

    @staticmethod
    def forward(ctx, x, k, stride, pad):
        _chk(x)
        x = x.contiguous().float()
        n, H, W, Cc = x.shape
        Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        out = torch.empty((n, Ho, Wo, Cc), dtype=torch.float32, device=x.device)
        out16 = torch.empty((n, Ho, Wo, Cc), dtype=ops.compute_dtype(), device=x.device)     # operand of the next conv
        idx = torch.empty((n, Ho, Wo, Cc), dtype=torch.uint8, device=x.device)
        check(lib.eoe_maxpool_fwd(_p(x), _p(out), _p(out16), _p(idx), n, H, W, Cc, k, stride, pad, dtype_code(out16.dtype),
                                  _stream()), "eoe_maxpool_fwd")
        ctx.save_for_backward(idx)
        ctx.geo = (n, H, W, Cc, k, stride, pad)
        ctx.mark_non_differentiable(out16)
        ctx.set_materialize_grads(False)       # else autograd zero-fills a gradient for the 16-bit copy every step
        return out, out16

    @staticmethod
    def backward(ctx, dout, _d16=None):
        (idx,) = ctx.saved_tensors
        n, H, W, Cc, k, stride, pad = ctx.geo
        dout = dout.contiguous().float()
        dx = torch.empty((n, H, W, Cc), dtype=torch.float32, device=dout.device)
        check(lib.eoe_maxpool_bwd(_p(dout), _p(idx), _p(dx), n, H, W, Cc, k, stride, pad, _stream()), "eoe_maxpool_bwd")
        return dx, None, None, None


class ChannelGateFunction(torch.autograd.Function):
    """x * sigmoid(mlp(avgpool x) + mlp(maxpool x)) (`cbam.py:31-66`); x fp32 NHWC"""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        _chk(x, w1, b1, w2, b2)
        x = x.contiguous().float()
        n, H, W, Cc = x.shape
        Ch = w1.shape[0]
        dev = x.device
        out = torch.empty_like(x)
        pooled = torch.empty((n, 2, Cc), dtype=torch.float32, device=dev)
        argmax = torch.empty((n, Cc), dtype=torch.int32, device=dev)
        hidden = torch.empty((n, 2, Ch), dtype=torch.float32, device=dev)
        scale = torch.empty((n, Cc), dtype=torch.float32, device=dev)
        w1c, b1c, w2c, b2c = (t.detach().contiguous() for t in (w1, b1, w2, b2))
        a = _lib.CGateArgs(_p(x), _p(out), _p(w1c), _p(b1c), _p(w2c), _p(b2c), _p(pooled), _p(argmax), _p(hidden), _p(scale),
                           n, H * W, Cc, Ch)
        check(lib.eoe_cgate_fwd(C.byref(a), _stream()), "eoe_cgate_fwd")
        ctx.save_for_backward(x, w1, b1, w2, b2, pooled, argmax, hidden, scale)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, w1, b1, w2, b2, pooled, argmax, hidden, scale = ctx.saved_tensors
        n, H, W, Cc = x.shape
        Ch = w1.shape[0]
        dev = x.device
        dout = dout.contiguous().float()
        dx = torch.empty_like(x)
        dscale = scratch("cg_dscale", (n, Cc), torch.float32, dev)
        dpooled = scratch("cg_dpooled", (n, 2, Cc), torch.float32, dev)
        dhidden = scratch("cg_dhidden", (n, 2, Ch), torch.float32, dev)
        dw1, db1, dw2, db2 = (_grad_target(t) for t in (w1, b1, w2, b2))
        w1c, b1c, w2c, b2c = (t.detach().contiguous() for t in (w1, b1, w2, b2))
        f = _lib.CGateArgs(_p(x), None, _p(w1c), _p(b1c), _p(w2c), _p(b2c), _p(pooled), _p(argmax), _p(hidden), _p(scale),
                           n, H * W, Cc, Ch)
        b = _lib.CGateBwdArgs(f, _p(dout), _p(dx), _p(dscale), _p(dpooled), _p(dhidden), _p(dw1), _p(db1), _p(dw2), _p(db2))
        check(lib.eoe_cgate_bwd(C.byref(b), _stream()), "eoe_cgate_bwd")
        return dx, dw1, db1, dw2, db2


class SpatialGateFunction(torch.autograd.Function):
    """x * sigmoid(bn(conv7x7([max_c x, mean_c x]))) (`cbam.py:76-92`); x fp32 NHWC; cfg = (training, eps, momentum)"""

    @staticmethod
    def forward(ctx, x, w, bn_w, bn_b, rm, rv, nbt, cfg):
        _chk(x, w, bn_w, bn_b, rm, rv)
        training, eps, momentum = cfg
        x = x.contiguous().float()
        n, H, W, Cc = x.shape
        dev = x.device
        out = torch.empty_like(x)
        comp = torch.empty((n, H, W, 2), dtype=torch.float32, device=dev)
        argmax = torch.empty((n, H, W), dtype=torch.int32, device=dev)
        z = torch.empty((n, H, W), dtype=torch.float32, device=dev)
        stats = torch.empty(2, dtype=torch.float32, device=dev)
        scale = torch.empty((n, H, W), dtype=torch.float32, device=dev)
        sums = scratch("sg_sums", (BN_SCRATCH,), torch.float32, dev)
        wc = w.detach().contiguous()
        a = _lib.SGateArgs(_p(x), _p(out), _p(wc), _p(bn_w), _p(bn_b), _p(rm), _p(rv), _p(nbt), _p(comp), _p(argmax), _p(z),
                           _p(stats), _p(scale), _p(sums), n, H, W, Cc, float(eps), float(momentum), 1 if training else 0,
                           None, None, 0)
        check(lib.eoe_sgate_fwd(C.byref(a), _stream()), "eoe_sgate_fwd")
        ctx.save_for_backward(x, w, bn_w, bn_b, comp, argmax, z, stats, scale)
        ctx.cfg = (training, float(eps), float(momentum))
        return out

    @staticmethod
    def backward(ctx, dout):
        x, w, bn_w, bn_b, comp, argmax, z, stats, scale = ctx.saved_tensors
        training, eps, momentum = ctx.cfg
        n, H, W, Cc = x.shape
        dev = x.device
        dout = dout.contiguous().float()
        dx = torch.empty_like(x)
        dscale = scratch("sg_dscale", (n, H, W), torch.float32, dev)
        dcomp = scratch("sg_dcomp", (n, H, W, 2), torch.float32, dev)
        red = scratch("sg_red", (SGATE_RED,), torch.float32, dev)
        wpart = scratch("sg_wpart", (n, 98), torch.float32, dev)
        sums = scratch("sg_sums", (BN_SCRATCH,), torch.float32, dev)
        dw = _grad_target(w)
        dg = _grad_target(bn_w) if bn_w is not None else None
        db = _grad_target(bn_b) if bn_b is not None else None
        wc = w.detach().contiguous()
        f = _lib.SGateArgs(_p(x), None, _p(wc), _p(bn_w), _p(bn_b), None, None, None, _p(comp), _p(argmax), _p(z), _p(stats),
                           _p(scale), _p(sums), n, H, W, Cc, eps, momentum, 1 if training else 0, None, None, 0)
        b = _lib.SGateBwdArgs(f, _p(dout), _p(dx), _p(dscale), _p(dcomp), _p(red), _p(dw), _p(dg), _p(db), _p(wpart))
        check(lib.eoe_sgate_bwd(C.byref(b), _stream()), "eoe_sgate_bwd")
        return dx, dw, dg, db, None, None, None, None


class SpatialGateAddReluFunction(torch.autograd.Function):
    """relu(x * sigmoid(bn(conv7x7([max_c x, mean_c x]))) + res): the spatial gate (`cbam.py:76-92`) and the residual junction
    of the BasicBlock (`resnet.py:143-147`) in one unit -- the gated tensor is never written.  Returns (out, 16-bit copy)."""

    @staticmethod
    def forward(ctx, x, res, w, bn_w, bn_b, rm, rv, nbt, cfg):
        _chk(x, res, w, bn_w, bn_b, rm, rv)
        training, eps, momentum = cfg
        x, res = x.contiguous().float(), res.contiguous().float()
        n, H, W, Cc = x.shape
        dev = x.device
        out = torch.empty_like(x)
        out16 = torch.empty(x.shape, dtype=ops.compute_dtype(), device=dev)
        comp = torch.empty((n, H, W, 2), dtype=torch.float32, device=dev)
        argmax = torch.empty((n, H, W), dtype=torch.int32, device=dev)
        z = torch.empty((n, H, W), dtype=torch.float32, device=dev)
        stats = torch.empty(2, dtype=torch.float32, device=dev)
        scale = torch.empty((n, H, W), dtype=torch.float32, device=dev)
        sums = scratch("sg_sums", (BN_SCRATCH,), torch.float32, dev)
        wc = w.detach().contiguous()
        a = _lib.SGateArgs(_p(x), _p(out), _p(wc), _p(bn_w), _p(bn_b), _p(rm), _p(rv), _p(nbt), _p(comp), _p(argmax), _p(z),
                           _p(stats), _p(scale), _p(sums), n, H, W, Cc, float(eps), float(momentum), 1 if training else 0,
                           _p(res), _p(out16), dtype_code(out16.dtype))
        check(lib.eoe_sgate_fwd(C.byref(a), _stream()), "eoe_sgate_fwd")
        ctx.save_for_backward(x, w, bn_w, bn_b, comp, argmax, z, stats, scale, out)
        ctx.cfg = (training, float(eps), float(momentum))
        ctx.mark_non_differentiable(out16)
        ctx.set_materialize_grads(False)       # else autograd zero-fills a gradient for the 16-bit copy every step
        return out, out16

    @staticmethod
    def backward(ctx, dout, _d16=None):
        x, w, bn_w, bn_b, comp, argmax, z, stats, scale, out = ctx.saved_tensors
        training, eps, momentum = ctx.cfg
        n, H, W, Cc = x.shape
        dev = x.device
        dout = dout.contiguous().float()
        g = torch.empty_like(out)                      # gradient at the junction = gradient of the residual branch
        check(lib.eoe_relu_bwd(_p(dout), _p(out), _p(g), out.numel(), _stream()), "eoe_relu_bwd")
        dx = torch.empty_like(x)
        dscale = scratch("sg_dscale", (n, H, W), torch.float32, dev)
        dcomp = scratch("sg_dcomp", (n, H, W, 2), torch.float32, dev)
        red = scratch("sg_red", (SGATE_RED,), torch.float32, dev)
        wpart = scratch("sg_wpart", (n, 98), torch.float32, dev)
        sums = scratch("sg_sums", (BN_SCRATCH,), torch.float32, dev)
        dw = _grad_target(w)
        dg = _grad_target(bn_w) if bn_w is not None else None
        db = _grad_target(bn_b) if bn_b is not None else None
        wc = w.detach().contiguous()
        f = _lib.SGateArgs(_p(x), None, _p(wc), _p(bn_w), _p(bn_b), None, None, None, _p(comp), _p(argmax), _p(z), _p(stats),
                           _p(scale), _p(sums), n, H, W, Cc, eps, momentum, 1 if training else 0, None, None, 0)
        b = _lib.SGateBwdArgs(f, _p(g), _p(dx), _p(dscale), _p(dcomp), _p(red), _p(dw), _p(dg), _p(db), _p(wpart))
        check(lib.eoe_sgate_bwd(C.byref(b), _stream()), "eoe_sgate_bwd")
        return dx, g, dw, dg, db, None, None, None, None


class CbamJunctionFunction(torch.autograd.Function):
    """relu(SpatialGate(ChannelGate(x)) + res): a BasicBlock's whole tail (`cbam.py:100-106`, `resnet.py:143-147`) as one unit
    (`eoe_cbam_junction_fwd / _bwd`): the channel-gated tensor and the spatial gate's input gradient are never written -- 22 + 32 bytes per
    activation element instead of the 30 + 44 of ChannelGateFunction + SpatialGateAddReluFunction.  Returns (out, 16-bit copy)."""

    @staticmethod
    def forward(ctx, x, res, w1, b1, w2, b2, w, bn_w, bn_b, rm, rv, nbt, cfg):
        _chk(x, res, w1, b1, w2, b2, w, bn_w, bn_b, rm, rv)
        training, eps, momentum = cfg
        x, res = x.contiguous().float(), res.contiguous().float()
        n, H, W, Cc = x.shape
        Ch = w1.shape[0]
        dev = x.device
        out = torch.empty_like(x)
        out16 = torch.empty(x.shape, dtype=ops.compute_dtype(), device=dev)
        pooled = torch.empty((n, 2, Cc), dtype=torch.float32, device=dev)
        argmax_c = torch.empty((n, Cc), dtype=torch.int32, device=dev)
        hidden = torch.empty((n, 2, Ch), dtype=torch.float32, device=dev)
        sc = torch.empty((n, Cc), dtype=torch.float32, device=dev)
        comp = torch.empty((n, H, W, 2), dtype=torch.float32, device=dev)
        argmax_p = torch.empty((n, H, W), dtype=torch.int32, device=dev)
        z = torch.empty((n, H, W), dtype=torch.float32, device=dev)
        stats = torch.empty(2, dtype=torch.float32, device=dev)
        sp = torch.empty((n, H, W), dtype=torch.float32, device=dev)
        sums = scratch("sg_sums", (BN_SCRATCH,), torch.float32, dev)
        w1c, b1c, w2c, b2c, wc = (t.detach().contiguous() for t in (w1, b1, w2, b2, w))
        cg = _lib.CGateArgs(_p(x), None, _p(w1c), _p(b1c), _p(w2c), _p(b2c), _p(pooled), _p(argmax_c), _p(hidden), _p(sc), n, H * W, Cc, Ch)
        sg = _lib.SGateArgs(None, _p(out), _p(wc), _p(bn_w), _p(bn_b), _p(rm), _p(rv), _p(nbt), _p(comp), _p(argmax_p), _p(z),
                            _p(stats), _p(sp), _p(sums), n, H, W, Cc, float(eps), float(momentum), 1 if training else 0,
                            _p(res), _p(out16), dtype_code(out16.dtype))
        check(lib.eoe_cbam_junction_fwd(C.byref(cg), C.byref(sg), _stream()), "eoe_cbam_junction_fwd")
        ctx.save_for_backward(x, w1, b1, w2, b2, w, bn_w, bn_b, pooled, argmax_c, hidden, sc, comp, argmax_p, z, stats, sp, out)
        ctx.cfg = (training, float(eps), float(momentum))
        ctx.mark_non_differentiable(out16)
        ctx.set_materialize_grads(False)       # else autograd zero-fills a gradient for the 16-bit copy every step
        return out, out16

    @staticmethod
    def backward(ctx, dout, _d16=None):
        x, w1, b1, w2, b2, w, bn_w, bn_b, pooled, argmax_c, hidden, sc, comp, argmax_p, z, stats, sp, out = ctx.saved_tensors
        training, eps, momentum = ctx.cfg
        n, H, W, Cc = x.shape
        Ch = w1.shape[0]
        dev = x.device
        dout = dout.contiguous().float()
        g = torch.empty_like(out)                      # gradient at the junction = gradient of the residual branch
        dx = torch.empty_like(x)
        dsc = scratch("cg_dscale8", (8, n, Cc), torch.float32, dev)               # EOE_CBAM_DSCALE_SLICES partial slices
        dpooled = scratch("cg_dpooled", (n, 2, Cc), torch.float32, dev)
        dhidden = scratch("cg_dhidden", (n, 2, Ch), torch.float32, dev)
        dsp = scratch("sg_dscale", (n, H, W), torch.float32, dev)
        dcomp = scratch("sg_dcomp4", (n, H, W, 4), torch.float32, dev)           # the unit's per-pixel record: 4 floats
        red = scratch("sg_red", (SGATE_RED,), torch.float32, dev)
        wpart = scratch("sg_wpart", (n, 98), torch.float32, dev)
        sums = scratch("sg_sums", (BN_SCRATCH,), torch.float32, dev)
        dw1, db1, dw2, db2, dw = (_grad_target(t) for t in (w1, b1, w2, b2, w))
        dg = _grad_target(bn_w) if bn_w is not None else None
        db = _grad_target(bn_b) if bn_b is not None else None
        w1c, b1c, w2c, b2c, wc = (t.detach().contiguous() for t in (w1, b1, w2, b2, w))
        cg = _lib.CGateArgs(_p(x), None, _p(w1c), _p(b1c), _p(w2c), _p(b2c), _p(pooled), _p(argmax_c), _p(hidden), _p(sc), n, H * W, Cc, Ch)
        cb = _lib.CGateBwdArgs(cg, None, _p(dx), _p(dsc), _p(dpooled), _p(dhidden), _p(dw1), _p(db1), _p(dw2), _p(db2))
        sg = _lib.SGateArgs(None, None, _p(wc), _p(bn_w), _p(bn_b), None, None, None, _p(comp), _p(argmax_p), _p(z), _p(stats),
                            _p(sp), _p(sums), n, H, W, Cc, eps, momentum, 1 if training else 0, None, None, 0)
        sb = _lib.SGateBwdArgs(sg, None, None, _p(dsp), _p(dcomp), _p(red), _p(dw), _p(dg), _p(db), _p(wpart))
        check(lib.eoe_cbam_junction_bwd(C.byref(cb), C.byref(sb), _p(dout), _p(out), _p(g), _stream()), "eoe_cbam_junction_bwd")
        return dx, g, dw1, db1, dw2, db2, dw, dg, db, None, None, None, None


def cbam_junction(x, res, cbam, training):
    """relu(cbam(x) + res) through the fused unit (ChannelGate MLP = cbam.ChannelGate.mlp[1] / [3], SpatialGate = cbam.SpatialGate.spatial)"""
    l1, l3 = cbam.ChannelGate.mlp[1], cbam.ChannelGate.mlp[3]
    sg = cbam.SpatialGate.spatial
    bn = sg.bn
    out, out16 = CbamJunctionFunction.apply(x, res, l1.weight, l1.bias, l3.weight, l3.bias, sg.conv.weight, bn.weight, bn.bias,
                                            bn.running_mean, bn.running_var, bn.num_batches_tracked, (training, bn.eps, bn.momentum))
    out._eoe16 = out16
    return out


def spatial_gate_add_relu(x, res, conv_w, bn, training):
    out, out16 = SpatialGateAddReluFunction.apply(x, res, conv_w, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                                  bn.num_batches_tracked, (training, bn.eps, bn.momentum))
    out._eoe16 = out16
    return out


class AddReluFunction(torch.autograd.Function):
    """relu(a + b): the residual junction of a BasicBlock (`resnet.py:146-147`)"""

    @staticmethod
    def forward(ctx, a, b):
        _chk(a, b)
        a, b = a.contiguous().float(), b.contiguous().float()
        out = torch.empty_like(a)
        out16 = torch.empty(a.shape, dtype=ops.compute_dtype(), device=a.device)              # operand of the next conv
        check(lib.eoe_add_relu_fwd(_p(a), _p(b), _p(out), _p(out16), dtype_code(out16.dtype), a.numel(), _stream()),
              "eoe_add_relu_fwd")
        ctx.save_for_backward(out)
        ctx.mark_non_differentiable(out16)
        ctx.set_materialize_grads(False)       # else autograd zero-fills a gradient for the 16-bit copy every step
        return out, out16

    @staticmethod
    def backward(ctx, dout, _d16=None):
        (out,) = ctx.saved_tensors
        dout = dout.contiguous().float()
        g = torch.empty_like(out)
        check(lib.eoe_relu_bwd(_p(dout), _p(out), _p(g), out.numel(), _stream()), "eoe_relu_bwd")
        # Both operands receive the SAME buffer (the junction's gradient is identical for the branch and for the shortcut).  The
        # shortcut's copy reaches ConvBnActPoolFunction.backward of the block's first convolution as `d_pass`, which ACCUMULATES
        # its dgrad onto it in place.  That is safe by graph order, not by luck of stream order: that node also needs the gradient
        # of its main output, which only exists after every consumer of the branch copy (conv2 / CBAM backward) has run, so the
        # branch copy is dead by then.  A tensor hook or retain_grad() on the junction output that keeps `g` alive would see the
        # accumulated values -- the modules of this package install none.
        return g, g.view_as(g)


class GlobalAvgPoolFunction(torch.autograd.Function):
    """nn.AvgPool2d(7) on the final 7x7 grid + flatten (`resnet.py:38,104-105`): fp32 NHWC -> [n, C]"""

    @staticmethod
    def forward(ctx, x):
        _chk(x)
        x = x.contiguous().float()
        n, H, W, Cc = x.shape
        pooled = torch.empty((n, 2, Cc), dtype=torch.float32, device=x.device)
        argmax = scratch("gap_arg", (n, Cc), torch.int32, x.device)
        check(lib.eoe_avgpool_fwd(_p(x), _p(pooled), _p(argmax), n, H * W, Cc, _stream()), "eoe_avgpool_fwd")
        ctx.shape = (n, H, W, Cc)
        return pooled[:, 0, :]

    @staticmethod
    def backward(ctx, dout):
        n, H, W, Cc = ctx.shape
        dout = dout.contiguous().float()
        dx = torch.empty((n, H, W, Cc), dtype=torch.float32, device=dout.device)
        check(lib.eoe_avgpool_bwd(_p(dout), _p(dx), n, H * W, Cc, _stream()), "eoe_avgpool_bwd")
        return dx


def max_pool(x, k, stride, pad):
    out, out16 = MaxPoolFunction.apply(x, k, stride, pad)
    out._eoe16 = out16
    return out


def add_relu(a, b):
    out, out16 = AddReluFunction.apply(a, b)
    out._eoe16 = out16
    return out
