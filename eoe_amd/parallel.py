"""Data parallelism for the hot path: one process per GPU, `torch.distributed` backend "nccl" (= RCCL over xGMI).

New relative to the reference, which is single-device (`src/eoe/main/__init__.py:110-114`); semantics as
SURVEY.md section 8e: every rank holds full weights and optimiser state, takes rows [r*B/R, (r+1)*B/R) of the
normal half and of the OE half of each step batch, computes  sum(local per-sample losses) / GLOBAL batch size,
and the parameter gradients are summed across ranks.  Gradients live in one flat fp32 arena in parameter order;
each ViT block's slice (28 MB for ViT-B/32) is one bucket whose all-reduce is issued from inside that block's
backward (RCCL runs it on its own stream, overlapped with the remaining backward kernels); the embedding /
head / remaining parameters form the last bucket.
"""
import os
import weakref
from typing import List, Optional

import torch
import torch.distributed as dist

from . import ops


def init_from_env(backend: str = "nccl"):
    """rank / world from the torchrun environment; returns (rank, world, local_rank)"""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_rows(n_normal: int, n_oe: int, rank: int, world: int) -> torch.Tensor:
    """global row indices of this rank's share of a step batch laid out as [normal half | OE half]
    (`bases.py:597`); keeps every local batch balanced"""
    a, b = (n_normal * rank) // world, (n_normal * (rank + 1)) // world
    c, d = (n_oe * rank) // world, (n_oe * (rank + 1)) // world
    return torch.cat([torch.arange(a, b), n_normal + torch.arange(c, d)])


class GradArena:
    """flat fp32 gradient arena over the trainable parameters of `model` (+ bucketed, overlapped all-reduce)"""

    def __init__(self, model: torch.nn.Module, process_group=None):
        self.model = model
        self.group = process_group
        params = [p for p in model.parameters() if p.requires_grad]
        if not params:
            raise ValueError("no trainable parameters")
        dev = params[0].device
        self.params = params
        self.offsets = {}
        tot = 0
        for p in params:
            self.offsets[id(p)] = tot
            tot += (p.numel() + 63) // 64 * 64          # 256-B aligned slices
        self.flat = torch.zeros(tot, dtype=torch.float32, device=dev)
        for p in params:
            o = self.offsets[id(p)]
            p._eoe_grad_buf = self.flat[o:o + p.numel()].view(p.shape)
        # buckets: one per module that owns a fused backward (ViT blocks), the rest in a final bucket
        self.block_buckets = []
        covered = set()
        for mod in model.modules():
            if hasattr(mod, "forward_tokens") and hasattr(mod, "_params"):
                ps = [p for p in mod._params() if p.requires_grad]
                if not ps:
                    continue
                lo = min(self.offsets[id(p)] for p in ps)
                hi = max(self.offsets[id(p)] + (p.numel() + 63) // 64 * 64 for p in ps)
                if hi - lo != sum((p.numel() + 63) // 64 * 64 for p in ps):
                    continue                              # not contiguous in the arena: leave to the final bucket
                self.block_buckets.append((ps[0], lo, hi))
                covered.update(id(p) for p in ps)
        self.rest = [p for p in params if id(p) not in covered]
        self.handles: List = []
        self._installed = False

    # -- overlap: called from VitBlockFunction.backward right after the block's kernels were enqueued
    def install_hooks(self):
        for first, lo, hi in self.block_buckets:
            ops.grad_ready_hooks[id(first)] = (weakref.ref(first), (lambda lo=lo, hi=hi: self._reduce_slice(lo, hi)))
        self._installed = True

    def remove_hooks(self):
        for first, _, _ in self.block_buckets:
            ops.grad_ready_hooks.pop(id(first), None)
        self._installed = False

    def _reduce_slice(self, lo, hi):
        if dist.is_initialized() and dist.get_world_size(self.group) > 1:
            self.handles.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        """all-reduce whatever was not reduced from inside backward, then wait for every bucket.  Gradients that
        autograd did not place in the arena (p.grad is not the arena view) are reduced individually."""
        if not (dist.is_initialized() and dist.get_world_size(self.group) > 1):
            self.handles.clear()
            return
        if self._installed:
            todo = self.rest
        else:
            todo = self.params
        stray = [p for p in todo if p.grad is not None and p.grad.data_ptr() != p._eoe_grad_buf.data_ptr()]
        inarena = [p for p in todo if p.grad is not None and p.grad.data_ptr() == p._eoe_grad_buf.data_ptr()]
        if inarena:
            if not self._installed:
                self.handles.append(dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            else:
                # the parameters outside the per-block buckets, as maximal contiguous runs of the arena: two collectives for the
                # ViT (embedding / ln_pre in front of the blocks, ln_post / proj / head behind them), one for the CNNs -- not one
                # small all-reduce per parameter
                spans = sorted((self.offsets[id(p)], self.offsets[id(p)] + (p.numel() + 63) // 64 * 64) for p in inarena)
                runs = [list(spans[0])]
                for lo, hi in spans[1:]:
                    if lo == runs[-1][1]:
                        runs[-1][1] = hi
                    else:
                        runs.append([lo, hi])
                for lo, hi in runs:
                    self.handles.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for p in stray:
            self.handles.append(dist.all_reduce(p.grad, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for h in self.handles:
            h.wait()
        self.handles.clear()


def all_gather_1d(t: torch.Tensor, group=None) -> torch.Tensor:
    """concatenate equal-length 1-D tensors from all ranks (scores / labels for the epoch AUC)"""
    if not (dist.is_initialized() and dist.get_world_size(group) > 1):
        return t
    out = [torch.empty_like(t) for _ in range(dist.get_world_size(group))]
    dist.all_gather(out, t.contiguous(), group=group)
    return torch.cat(out)
