"""Fused multi-tensor Adam: one kernel launch per step over a chunk table (C ABI `eoe_adam_multi`).

Replaces `torch.optim.Adam(model.parameters(), lr=lr, weight_decay=wdk, amsgrad=False)` constructed inside the
reference's `train_cls` (`src/eoe/training/ad_trainer.py:383`) with the same `torch.optim.Optimizer` API
(`zero_grad / step / state_dict / load_state_dict / param_groups`, so `MultiStepLR` (:384) drives it unchanged)
and the same arithmetic: L2-in-gradient weight decay, bias correction, eps 1e-8; parameters whose grad is None
(frozen by `freeze_parts`, `ad_trainer.py:593-596`) are skipped and their step count does not advance.
"""
import ctypes as C
import math
import os

import numpy as np
import torch

from . import _lib, ops
from ._lib import lib, check

_CHUNK_DT = np.dtype([("p_off", "<i8"), ("g_off", "<i8"), ("m_off", "<i8"), ("v_off", "<i8"), ("n", "<i4"),
                      ("group", "<i4")])
assert _CHUNK_DT.itemsize == C.sizeof(_lib.AdamChunk)
_TILE_DT = np.dtype([("p_off", "<i8"), ("g_off", "<i8"), ("m_off", "<i8"), ("v_off", "<i8"), ("d16", "<u8"), ("d16_t", "<u8"),
                     ("rows", "<i4"), ("cols", "<i4"), ("tile", "<i4"), ("group", "<i4")])
assert _TILE_DT.itemsize == C.sizeof(_lib.AdamTile)
# 2-D weights whose 16-bit operand copies exist are updated by 64 x 64 tiles and get the copies rewritten in the same pass
# (eoe_adam_tiles, include/eoe_hip.h); EOE_ADAM_TILES=0: every parameter through the chunk kernel, the copies by the cast pass (A/B, tests)
ADAM_TILES = os.environ.get("EOE_ADAM_TILES", "1") != "0"


_warned_unscaled = False


def _warn_if_unscaled_fp16():
    """once per process: fp16 compute with the gradient scale left at 1 (ops.set_grad_scale)"""
    global _warned_unscaled
    if not _warned_unscaled and ops.compute_dtype() == torch.float16 and ops.grad_scale() == 1.0:
        _warned_unscaled = True
        import warnings
        warnings.warn("eoe_amd: fp16 compute without a gradient scale -- the 16-bit backward chain underflows once gradients get small "
                      "(the 12-layer ViT leaves the fp32 trajectory within 20 steps); call eoe_amd.set_grad_scale("
                      "eoe_amd.default_grad_scale()) before training, as eoe_amd.training and bench.py do", stacklevel=3)


class _NonFiniteGuard:
    """Skip-on-overflow for the scaled fp16 step, without a host synchronisation (C ABI `eoe_grads_nonfinite`).

    The reference's numerical-failure policy is the per-epoch NaN check with retry (`ad_trainer.py:257-280, 448-449`).  This build
    multiplies the loss gradient by a scale so that the 16-bit backward chain does not underflow (ops.set_grad_scale); a scale can
    also overflow that chain.  With the guard on (default whenever the gradient scale is not 1) the optimiser first streams over the
    gradients it is about to apply; if any is inf / NaN the update kernels of that step leave parameters and moments untouched (a
    device flag: the host is not asked).  `skipped_steps()` reads the device counter (one small copy) and takes the skipped steps out
    of the per-parameter step counts again, so later bias corrections are those of the steps actually applied; trainers poll it
    every few steps to halve the scale after an overflow and to grow it back after a run of clean steps (training/ad_trainer.py)."""

    def _guard_init(self, guard):
        self.guard = guard                    # None: on iff the gradient scale is not 1; True / False: forced
        self._guard_state = None              # device int32[4]: flag even, flag odd, skipped, checked
        self._guard_parity = 0
        self._guard_seen = 0                  # skipped steps already reported / taken out of the step counts
        self._guard_counted = {}              # id -> parameter whose step count advanced on the steps not yet polled

    def _guard_active(self) -> bool:
        return (ops.grad_scale() != 1.0) if self.guard is None else bool(self.guard)

    def _guard_begin(self, device):
        """one call per optimiser step, before the first group: returns (state tensor, parity)"""
        if self._guard_state is None or self._guard_state.device != device:
            self._guard_state = torch.zeros(4, dtype=torch.int32, device=device)
        self._guard_parity ^= 1
        return self._guard_state, self._guard_parity

    def skipped_steps(self) -> int:
        """number of optimiser steps dropped for non-finite gradients since the last call (synchronises with the device once).
        The step counts that advanced on those steps are taken back."""
        if self._guard_state is None:
            return 0
        st = self._guard_state.cpu().tolist()
        total = st[2] + (1 if st[self._guard_parity] else 0)      # the newest step's flag has not been retired into the count yet
        new = total - self._guard_seen
        self._guard_seen = total
        if new > 0:
            for p in self._guard_counted.values():
                st_p = self.state.get(p)
                if st_p is not None and "step" in st_p:
                    st_p["step"] -= float(new)
        self._guard_counted = {}
        return new


class FusedAdam(_NonFiniteGuard, torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False, guard=None):
        if amsgrad:
            raise NotImplementedError("amsgrad is not used by the reference trainer (ad_trainer.py:383)")
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False)
        super().__init__(params, defaults)
        self._tables = {}
        self._step_flats = []
        self._guard_init(guard)

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._step_flats = []                  # the loaded "step" tensors are their own storage: per-parameter path from here on
        self._tables = {}

    def _advance_steps(self, active):
        """step counts of `active` after this step's increment (ints)"""
        ids = tuple(id(p) for p in active)
        for flat, owners in self._step_flats:
            if owners == ids and all(self.state[p]["step"].data_ptr() == flat.data_ptr() + 4 * i for i, p in enumerate(active)):
                flat += 1
                return [int(round(v)) for v in flat.tolist()]
        steps = []
        for p in active:
            st = self.state[p]
            st["step"] += 1
            steps.append(int(round(st["step"].item())))
        return steps

    def _init_state(self, group):
        """exp_avg / exp_avg_sq of a whole group live in two flat arenas (one allocation each)"""
        need = [p for p in group["params"] if p.requires_grad and "exp_avg" not in self.state[p]]
        if not need:
            return
        tot = sum((p.numel() + 3) // 4 * 4 for p in need)
        dev = need[0].device
        m_arena = torch.zeros(tot, dtype=torch.float32, device=dev)
        v_arena = torch.zeros(tot, dtype=torch.float32, device=dev)
        off = 0
        # the step counts of the group live in ONE host tensor, each state's "step" a 0-dim view of it (what torch's state_dict expects per
        # parameter): the per-step increment and read-back are two tensor operations instead of two per parameter (0.5 ms of host time per step)
        steps_flat = torch.zeros(len(need), dtype=torch.float32)
        self._step_flats.append((steps_flat, tuple(id(p) for p in need)))
        for i, p in enumerate(need):
            n = p.numel()
            st = self.state[p]
            st["step"] = steps_flat[i]
            st["exp_avg"] = m_arena[off:off + n].view(p.shape)
            st["exp_avg_sq"] = v_arena[off:off + n].view(p.shape)
            off += (n + 3) // 4 * 4

    def _table(self, gi, active, steps):
        """device chunk table for the active parameters of group gi, cached on pointers and step grouping"""
        distinct = sorted(set(steps))
        sig = (tuple((p.data_ptr(), p.grad.data_ptr(), self.state[p]["exp_avg"].data_ptr(), p.numel()) for p in active),
               tuple(distinct.index(s) for s in steps))
        hit = self._tables.get(gi)
        if hit is not None and hit[0] == sig:
            return hit[1:], distinct
        # bases rounded DOWN to 16 bytes: the kernel takes its float4 path for a chunk iff all four element offsets are multiples
        # of 4, which is a statement about the ADDRESS only if the bases themselves are 16-byte aligned (a parameter that is a
        # 4-byte-aligned view of a user buffer may be the lowest address)
        pb = min(p.data_ptr() for p in active) & ~15
        gb = min(p.grad.data_ptr() for p in active) & ~15
        mb = min(self.state[p]["exp_avg"].data_ptr() for p in active) & ~15
        vb = min(self.state[p]["exp_avg_sq"].data_ptr() for p in active) & ~15
        rows = []
        for p, s in zip(active, steps):
            st = self.state[p]
            po, go = (p.data_ptr() - pb) // 4, (p.grad.data_ptr() - gb) // 4
            mo, vo = (st["exp_avg"].data_ptr() - mb) // 4, (st["exp_avg_sq"].data_ptr() - vb) // 4
            n = p.numel()
            for c0 in range(0, n, _lib.ADAM_CHUNK):
                rows.append((po + c0, go + c0, mo + c0, vo + c0, min(_lib.ADAM_CHUNK, n - c0), distinct.index(s)))
        arr = np.array(rows, dtype=_CHUNK_DT)
        tab = torch.from_numpy(arr.view(np.uint8).copy()).to(active[0].device)
        self._tables[gi] = (sig, tab, len(rows), (pb, gb, mb, vb))
        return (tab, len(rows), (pb, gb, mb, vb)), distinct

    def _split(self, gi, active, steps, distinct, bases, pairs):
        """the group's table in two parts: 64 x 64 tile rows for the parameters in `pairs` (id -> (copy, transposed copy)), chunk rows for
        the rest; same bases as the full table (which the non-finite check keeps walking).  Cached on the full table's identity and the
        copies' addresses."""
        full = self._tables[gi]
        # (keyed on the full table's own signature -- the addresses it was built from -- not on id() of its tensor: CPython reuses ids)
        sig = (full[0], tuple((k, d.data_ptr(), dt.data_ptr()) for k, (d, dt) in pairs.items()))
        hit = self._tables.get(("split", gi))
        if hit is not None and hit[0] == sig:
            return hit[1:]
        pb, gb, mb, vb = bases
        rows, tiles = [], []
        for p, s in zip(active, steps):
            st = self.state[p]
            po, go = (p.data_ptr() - pb) // 4, (p.grad.data_ptr() - gb) // 4
            mo, vo = (st["exp_avg"].data_ptr() - mb) // 4, (st["exp_avg_sq"].data_ptr() - vb) // 4
            pr = pairs.get(id(p))
            if pr is None:
                n = p.numel()
                for c0 in range(0, n, _lib.ADAM_CHUNK):
                    rows.append((po + c0, go + c0, mo + c0, vo + c0, min(_lib.ADAM_CHUNK, n - c0), distinct.index(s)))
            else:
                R, Cc = p.shape
                for t in range(((R + 63) // 64) * ((Cc + 63) // 64)):
                    tiles.append((po, go, mo, vo, pr[0].data_ptr(), pr[1].data_ptr(), R, Cc, t, distinct.index(s)))
        dev = active[0].device
        ctab = torch.from_numpy(np.array(rows, dtype=_CHUNK_DT).view(np.uint8).copy()).to(dev) if rows else None
        ttab = torch.from_numpy(np.array(tiles, dtype=_TILE_DT).view(np.uint8).copy()).to(dev)
        self._tables[("split", gi)] = (sig, ctab, len(rows), ttab, len(tiles))
        return ctab, len(rows), ttab, len(tiles)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        work, work_index = [], {}
        for gi, group in enumerate(self.param_groups):
            active = [p for p in group["params"] if p.grad is not None]
            if not active:
                continue
            for p in active:
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous() or not p.grad.is_contiguous():
                    raise RuntimeError("FusedAdam needs contiguous fp32 parameters and gradients on the GPU")
                if getattr(p, "_eoe_fp16_weight", False):
                    raise NotImplementedError("fp16-weights mode (eoe_amd.models.convert_weights) is the CLIP objective's: the reference "
                                              "trains such models with SGD (ad_trainer.py:380-381) -- use eoe_amd.FusedSGD")
            self._init_state(group)
            steps = self._advance_steps(active)
            (tab, n_chunks, bases), distinct = self._table(gi, active, steps)
            if len(distinct) > _lib.ADAM_GROUPS:
                raise RuntimeError("FusedAdam: more than %d distinct step counts in one group" % _lib.ADAM_GROUPS)
            work.append((group, active, tab, n_chunks, bases, distinct))
            work_index[id(group)] = (gi, steps)
        if not work:
            return loss
        stream = torch.cuda.current_stream().cuda_stream
        skip = None
        if self._guard_active():
            # every group is checked before any group is updated: a step is dropped whole or applied whole
            state, parity = self._guard_begin(work[0][1][0].device)
            for k, (group, active, tab, n_chunks, bases, distinct) in enumerate(work):
                check(lib.eoe_grads_nonfinite(bases[1], tab.data_ptr(), n_chunks, state.data_ptr(), parity, 1 if k == 0 else 0, stream),
                      "eoe_grads_nonfinite")
                self._guard_counted.update((id(p), p) for p in active)
            skip = state.data_ptr() + 4 * parity
        _warn_if_unscaled_fp16()
        for group, active, tab, n_chunks, bases, distinct in work:
            beta1, beta2 = group["betas"]
            lr = float(group["lr"])
            sc = _lib.AdamScalars()
            sc.grad_scale_inv = 1.0 / ops.grad_scale()       # the losses' backward multiplied every gradient by the scale
            for i, s in enumerate(distinct):
                bc1 = 1.0 - beta1 ** s
                bc2 = 1.0 - beta2 ** s
                sc.step_size[i] = lr / bc1
                sc.bc2_sqrt[i] = math.sqrt(bc2)
            # 2-D weights with live 16-bit operand copies: by tiles, the copies rewritten from the update's registers
            pairs = {}
            if ADAM_TILES and not torch.cuda.is_current_stream_capturing():
                for p in active:
                    if p.dim() != 2 or p.shape[0] % 4 or p.shape[1] % 4:
                        continue
                    st = self.state[p]
                    # the tile kernel's 16-byte accesses need what the chunk kernel tests per chunk: all four element offsets multiples of 4
                    if all(((t.data_ptr() - b) // 4) % 4 == 0 for t, b in zip((p, p.grad, st["exp_avg"], st["exp_avg_sq"]), bases)):
                        pr = ops.shadow.entry(p)
                        if pr is not None:
                            pairs[id(p)] = pr
            hyper = (float(beta1), float(beta2), float(group["eps"]), float(group["weight_decay"]))
            if pairs:
                gi, steps = work_index[id(group)]
                ctab, n_c, ttab, n_t = self._split(gi, active, steps, distinct, bases, pairs)
                if n_c:
                    check(lib.eoe_adam_multi(bases[0], bases[1], bases[2], bases[3], ctab.data_ptr(), n_c, C.byref(sc), *hyper,
                                             None, _lib.EOE_BF16, skip, stream), "eoe_adam_multi")
                check(lib.eoe_adam_tiles(bases[0], bases[1], bases[2], bases[3], ttab.data_ptr(), n_t, C.byref(sc), *hyper,
                                         ops.dtype_code(ops._compute_dtype), skip, stream), "eoe_adam_tiles")
            else:
                check(lib.eoe_adam_multi(bases[0], bases[1], bases[2], bases[3], tab.data_ptr(), n_chunks, C.byref(sc), *hyper,
                                         None, _lib.EOE_BF16, skip, stream), "eoe_adam_multi")
            torch._C._increment_version(active)     # the kernel wrote in place: invalidate 16-bit weight copies ...
            for p in active:                        # ... except the ones it has just rewritten
                pr = pairs.get(id(p))
                if pr is not None:
                    ops.shadow.mark(p, pr[0], pr[1])
        return loss


def is_fp16_weight(p) -> bool:
    """True for a parameter `eoe_amd.models.convert_weights` marked: an fp16 tensor in the reference (clip/model.py:371-392)"""
    return bool(getattr(p, "_eoe_fp16_weight", False))


class FusedSGD(_NonFiniteGuard, torch.optim.Optimizer):
    """`torch.optim.SGD(params, lr, momentum=0.9, nesterov=True, weight_decay=wdk)` as the reference constructs it for CLIP models
    (`ad_trainer.py:380-381`), one kernel per step (`eoe_sgd_multi`): same Optimizer API, dampening 0, L2-in-gradient weight decay,
    momentum buffers created at the first step (zero-initialised arena: `buf = momentum * 0 + g` is torch's first-step `buf = g`).
    Parameters marked by `eoe_amd.models.convert_weights` (the reference's fp16-weights mode: `build_model` converts CLIP's convolution /
    linear / attention / projection parameters to fp16, `clip/model.py:371-392, 430`, and SGD then updates fp16 tensors) are updated
    with torch's fp16 arithmetic -- every op of the update rounded to fp16 -- on fp32 storage that keeps fp16-representable values."""

    def __init__(self, params, lr=1e-3, momentum=0.0, weight_decay=0.0, nesterov=False, dampening=0.0, guard=None):
        if dampening != 0.0:
            raise NotImplementedError("dampening is not used by the reference trainer (ad_trainer.py:381)")
        if nesterov and momentum <= 0:
            raise ValueError("Nesterov momentum requires a momentum")
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay, nesterov=nesterov, dampening=0.0))
        self._tables = {}
        self._guard_init(guard)

    def skipped_steps(self) -> int:
        """SGD keeps no step counts: only the counter is read"""
        self._guard_counted = {}
        return super().skipped_steps()

    def _init_state(self, group):
        need = [p for p in group["params"] if p.grad is not None and "momentum_buffer" not in self.state[p]]
        if not need:
            return
        tot = sum((p.numel() + 3) // 4 * 4 for p in need)
        arena = torch.zeros(tot, dtype=torch.float32, device=need[0].device)
        off = 0
        for p in need:
            n = p.numel()
            self.state[p]["momentum_buffer"] = arena[off:off + n].view(p.shape)
            off += (n + 3) // 4 * 4

    def _table(self, gi, active):
        sig = tuple((p.data_ptr(), p.grad.data_ptr(), self.state[p]["momentum_buffer"].data_ptr(), p.numel(), is_fp16_weight(p)) for p in active)
        hit = self._tables.get(gi)
        if hit is not None and hit[0] == sig:
            return hit[1:]
        pb = min(p.data_ptr() for p in active) & ~15          # 16-byte aligned bases (see FusedAdam._table)
        gb = min(p.grad.data_ptr() for p in active) & ~15
        mb = min(self.state[p]["momentum_buffer"].data_ptr() for p in active) & ~15
        rows = []
        for p in active:
            po, go = (p.data_ptr() - pb) // 4, (p.grad.data_ptr() - gb) // 4
            mo = (self.state[p]["momentum_buffer"].data_ptr() - mb) // 4
            n = p.numel()
            flag = _lib.CHUNK_FP16 if is_fp16_weight(p) else 0       # fp16-weights mode: torch's fp16 update, op by op (eoe_sgd_multi)
            for c0 in range(0, n, _lib.ADAM_CHUNK):
                rows.append((po + c0, go + c0, mo + c0, 0, min(_lib.ADAM_CHUNK, n - c0), flag))
        arr = np.array(rows, dtype=_CHUNK_DT)
        tab = torch.from_numpy(arr.view(np.uint8).copy()).to(active[0].device)
        self._tables[gi] = (sig, tab, len(rows), (pb, gb, mb))
        return tab, len(rows), (pb, gb, mb)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        work = []
        for gi, group in enumerate(self.param_groups):
            active = [p for p in group["params"] if p.grad is not None]
            if not active:
                continue
            for p in active:
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous() or not p.grad.is_contiguous():
                    raise RuntimeError("FusedSGD needs contiguous fp32 parameters and gradients on the GPU")
            self._init_state(group)
            work.append((group, active) + tuple(self._table(gi, active)))
        if not work:
            return loss
        stream = torch.cuda.current_stream().cuda_stream
        skip = None
        if self._guard_active():
            state, parity = self._guard_begin(work[0][1][0].device)
            for k, (group, active, tab, n_chunks, bases) in enumerate(work):
                check(lib.eoe_grads_nonfinite(bases[1], tab.data_ptr(), n_chunks, state.data_ptr(), parity, 1 if k == 0 else 0, stream),
                      "eoe_grads_nonfinite")
            skip = state.data_ptr() + 4 * parity
        _warn_if_unscaled_fp16()
        for group, active, tab, n_chunks, bases in work:
            check(lib.eoe_sgd_multi(bases[0], bases[1], bases[2], tab.data_ptr(), n_chunks, float(group["lr"]),
                                    float(group["momentum"]), float(group["weight_decay"]), 1 if group["nesterov"] else 0,
                                    1.0 / ops.grad_scale(), skip, stream), "eoe_sgd_multi")
            torch._C._increment_version(active)
        return loss
