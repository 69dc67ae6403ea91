"""Data parallelism for the hot path: one process per GPU, `torch.distributed` backend "nccl" (= RCCL over xGMI).

New relative to the reference, which is single-device (`src/eoe/main/__init__.py:110-114`); semantics as
SURVEY.md section 8e: every rank holds full weights and optimiser state, takes rows [r*B/R, (r+1)*B/R) of the
normal half and of the OE half of each step batch, computes  sum(local per-sample losses) / GLOBAL batch size,
and the parameter gradients are summed across ranks.  Gradients live in one flat fp32 arena in parameter order;
each ViT block's slice (28 MB for ViT-B/32) is one bucket whose all-reduce is issued from inside that block's
backward (RCCL runs it on its own stream, overlapped with the remaining backward kernels); all other parameters
(and every layer of the conv nets) go out in ~8 MB runs as their gradients arrive (`GradArena`).

BatchNorm encoders (CNN32 / CNN28 / WideResNet + CBAM): the reference is single-device, its BatchNorm layers see the whole
step batch.  `enable_sync_bn()` keeps that meaning under data parallelism: every training-mode BatchNorm reduction of the HIP
library -- forward (sum, sum of squares, rows), backward (sum g, sum g*xhat, rows) -- is summed over the ranks before it is used
(C ABI hook `eoe_set_bn_sync`), so activations, gradients and running statistics are those of the global batch whatever the
sharding (ragged shards included: the row count travels with the sums).  Without it the statistics are per rank
(`DistributedDataParallel`-without-SyncBatchNorm semantics).  The ViT headline model has no BatchNorm.
"""
import ctypes as C
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


class NativeComm:
    """the C-ABI communicator (`eoe_comm_*`, csrc/comm.cpp): RCCL behind libeoe_hip.so with its own side stream, for callers
    that drive the hot path without torch.distributed collectives.  The 128-byte RCCL id is made on rank 0 and handed to the
    other ranks through the torch.distributed process group that the launcher set up (any backend; world 1 needs none)."""

    def __init__(self, rank: int = None, world: int = None, device: int = None, algo: int = None):
        import ctypes as C
        from . import _lib
        self._lib, self._C = _lib, C
        if world is None:
            world = dist.get_world_size() if dist.is_initialized() else 1
            rank = dist.get_rank() if dist.is_initialized() else 0
        self.rank, self.world = rank, world
        self.device = torch.cuda.current_device() if device is None else device
        self.algo = _lib.EOE_COMM_ALGO_RING if algo is None else algo
        handle = C.c_void_p()
        _lib.check(_lib.lib.eoe_comm_init(self.fresh_id(), rank, world, self.device, C.byref(handle)), "eoe_comm_init")
        self.handle = handle

    def fresh_id(self) -> bytes:
        """a new RCCL id: drawn on rank 0 (`eoe_comm_unique_id`), the same bytes on every rank (broadcast over the launcher's process group)"""
        _lib, C = self._lib, self._C
        ident = torch.zeros(_lib.EOE_COMM_ID_BYTES, dtype=torch.uint8)
        if self.rank == 0:
            buf = (C.c_char * _lib.EOE_COMM_ID_BYTES)()
            _lib.check(_lib.lib.eoe_comm_unique_id(buf), "eoe_comm_unique_id")
            ident = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
        if self.world > 1:
            dev = torch.device("cuda", self.device) if dist.get_backend() == "nccl" else torch.device("cpu")
            ident = ident.to(dev)
            dist.broadcast(ident, src=0)
            ident = ident.cpu()
        return bytes(ident.numpy().tobytes())

    @staticmethod
    def _code(t: torch.Tensor) -> int:
        from . import _lib
        return {torch.float32: _lib.EOE_F32, torch.float16: _lib.EOE_F16, torch.bfloat16: _lib.EOE_BF16,
                torch.int64: _lib.EOE_COMM_I64}[t.dtype]

    def all_reduce_async(self, t: torch.Tensor):
        """SUM all-reduce of a contiguous device tensor, in place, on the communicator's side stream behind the current stream"""
        assert t.is_cuda and t.is_contiguous()
        self._lib.check(self._lib.lib.eoe_comm_allreduce_sum_async(self.handle, t.data_ptr(), t.numel(), self._code(t), self.algo,
                                                                   torch.cuda.current_stream().cuda_stream), "eoe_comm_allreduce_sum_async")

    def all_gather_async(self, t: torch.Tensor) -> torch.Tensor:
        assert t.is_cuda and t.is_contiguous()
        out = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        self._lib.check(self._lib.lib.eoe_comm_allgather_async(self.handle, t.data_ptr(), out.data_ptr(), t.numel(), self._code(t),
                                                               torch.cuda.current_stream().cuda_stream), "eoe_comm_allgather_async")
        return out

    def join(self):
        """the current stream waits for every collective issued so far"""
        self._lib.check(self._lib.lib.eoe_comm_join(self.handle, torch.cuda.current_stream().cuda_stream), "eoe_comm_join")

    def close(self):
        global _bn_sync_cb
        if self.handle is not None:
            if _bn_sync_cb is self:                       # the library clears its hook in eoe_comm_destroy; drop our reference too
                _bn_sync_cb = None
            self._lib.check(self._lib.lib.eoe_comm_destroy(self.handle), "eoe_comm_destroy")
            self.handle = None


def make_comm(prefer: str = "auto", algo: str = "rs_ag"):
    """the gradient transport of a data-parallel run: (NativeComm or None, description).  "auto" (or the environment's EOE_COMM)
    takes the C-ABI communicator -- RCCL on its own side HIP stream, reduce-scatter + all-gather per bucket, BatchNorm sums inside the
    library -- whenever the process group runs on RCCL ("nccl"); gloo groups (the CPU / one-GPU rehearsals: RCCL refuses two ranks on
    one device) keep torch.distributed collectives.  Every rank must end up on the same transport, so the outcome of the communicator's
    construction is agreed on with one MIN all-reduce."""
    from . import _lib
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return None, "none"
    prefer = os.environ.get("EOE_COMM", prefer)
    kind = f"torch.distributed ({dist.get_backend()})"
    if not (prefer == "native" or (prefer == "auto" and dist.get_backend() == "nccl")):
        return None, kind
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    ok, err, comm = torch.ones(1, device=dev), "", None
    try:
        comm = NativeComm(algo=_lib.EOE_COMM_ALGO_RS_AG if algo == "rs_ag" else _lib.EOE_COMM_ALGO_RING)
    except Exception as e:
        ok.zero_()
        err = repr(e)[:160]
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if ok.item() < 1:
        if comm is not None:
            comm.close()
        return None, kind + f" [native communicator failed on some rank: {err}]"
    return comm, f"eoe_comm (RCCL, side stream, {algo})"


class GradArena:
    """flat fp32 gradient arena over the trainable parameters of `model` + bucketed all-reduce overlapped with backward.

    Buckets are contiguous slices of the arena: one per module with a fused backward (a ViT block: 28 MB, its all-reduce is
    issued from inside `VitBlockFunction.backward` as soon as the block's kernels are enqueued), and the remaining parameters
    (embedding / head of the ViT; every layer of CNN32 / CNN28 / WideResNet) in runs of about `bucket_bytes`, each issued by
    `post_accumulate_grad` hooks when the last gradient of the run has arrived -- backward reaches the layers last to first,
    so the deep layers' gradients travel while the shallow layers are still being differentiated.  RCCL runs the collectives on
    its own stream behind an event of the compute stream.  The sequence of collectives is the same on every rank (it depends
    only on the autograd graph), which is what RCCL needs."""

    def __init__(self, model: torch.nn.Module, process_group=None, bucket_bytes: int = 8 << 20, comm: "NativeComm" = None,
                 bucket_dtype: Optional[torch.dtype] = None):
        self.model = model
        self.group = process_group
        self.comm = comm                                  # None: torch.distributed collectives; else the C-ABI communicator
        # optional (round 5): the buckets travel as 16-bit values -- half the bytes over xGMI (351 -> 175 MB per step for ViT-B/32).  A bucket's
        # slice is cast into a 16-bit twin of the arena, the twin is summed across the ranks, and finish() casts the sums back.  bfloat16 keeps
        # fp32's range (an fp16 run's gradients are scaled by 256: float16 buckets could overflow where the fp32 sum would not).  The sum of R
        # rounded addends is within (R + 1) half-ulps of the fp32 sum: 2e-3 relative per element at 8 ranks -- noise of zero mean that a
        # 1e-3 loss trajectory holds (tests/test_cpu_distributed.py); default None = fp32 buckets, the reference arithmetic
        assert bucket_dtype in (None, torch.float32, torch.bfloat16, torch.float16), bucket_dtype
        self.bucket_dtype = None if bucket_dtype in (None, torch.float32) else bucket_dtype
        params = [p for p in model.parameters() if p.requires_grad]
        if not params:
            raise ValueError("no trainable parameters")
        dev = params[0].device
        self.params = params
        self.offsets = {}
        tot = 0
        for p in params:
            self.offsets[id(p)] = tot
            tot += self._span(p)
        self.flat = torch.zeros(tot, dtype=torch.float32, device=dev)
        self.flat16 = torch.zeros(tot, dtype=self.bucket_dtype, device=dev) if self.bucket_dtype is not None else None
        for p in params:
            o = self.offsets[id(p)]
            p._eoe_grad_buf = self.flat[o:o + p.numel()].view(p.shape)
        # buckets of the fused blocks
        self.block_buckets = []
        covered = set()
        for mod in model.modules():
            if hasattr(mod, "forward_tokens") and hasattr(mod, "_params"):
                ps = [p for p in mod._params() if p.requires_grad]
                if not ps:
                    continue
                lo = min(self.offsets[id(p)] for p in ps)
                hi = max(self.offsets[id(p)] + self._span(p) for p in ps)
                if hi - lo != sum(self._span(p) for p in ps):
                    continue                              # not contiguous in the arena: leave to the generic buckets
                self.block_buckets.append((ps[0], lo, hi))
                covered.update(id(p) for p in ps)
        self.rest = [p for p in params if id(p) not in covered]
        # generic buckets over the rest: maximal contiguous runs of the arena, cut every `bucket_bytes`
        self.run_buckets = []                             # [params, lo, hi]
        cur, cur_lo, cur_hi = [], None, None
        for p in self.rest:                               # arena order = parameter order
            lo, hi = self.offsets[id(p)], self.offsets[id(p)] + self._span(p)
            if cur and (lo != cur_hi or (cur_hi - cur_lo) * 4 >= bucket_bytes):
                self.run_buckets.append((cur, cur_lo, cur_hi))
                cur = []
            if not cur:
                cur_lo = lo
            cur.append(p)
            cur_hi = hi
        if cur:
            self.run_buckets.append((cur, cur_lo, cur_hi))
        self.handles: List = []
        self._installed = False
        self._hook_handles = []
        self._pending = {}
        self.issued = []                                  # (lo, hi) of the collectives of the current step, in issue order
        self._sent16 = []

    @staticmethod
    def _span(p) -> int:
        return (p.numel() + 63) // 64 * 64                # 256-B aligned slices

    # -- overlap
    def install_hooks(self):
        """fused blocks: called from VitBlockFunction.backward right after the block's kernels were enqueued; everything else:
        post-accumulate hooks that count a run's gradients in"""
        for first, lo, hi in self.block_buckets:
            ops.grad_ready_hooks[id(first)] = (weakref.ref(first), (lambda lo=lo, hi=hi: self._reduce_slice(lo, hi)))
        for b, (ps, lo, hi) in enumerate(self.run_buckets):
            self._pending[b] = len(ps)
            for p in ps:
                self._hook_handles.append(p.register_post_accumulate_grad_hook(lambda p, b=b: self._arrived(p, b)))
        self._installed = True
        ops.async_wgrad_blockers += 1          # the hooks send a gradient the moment autograd has it: no weight gradient may still be in flight
        # With collectives in flight RCCL's workgroups occupy CUs for the length of a bucket, and the stream-K wgrad launch wants every CU
        # at once (one k-range per CU, each waiting for its predecessor's partial): ranges that find no CU would start a second round.
        # Data parallel runs keep the one-tile-per-workgroup launch (216 of 256 CUs) with LayerNorm-1 backward beside it instead.
        # (bit 1 is OR-ed into whatever the switch holds -- a user's A/B bits, another arena's request -- and remove_hooks puts back
        #  exactly what this arena found)
        self._tn_flags_before = None
        if self._world() > 1:
            from . import _lib
            self._tn_flags_before = _lib.get_option("tn_flags")
            _lib.set_option("tn_flags", self._tn_flags_before | 2)

    def _world(self) -> int:
        if self.comm is not None:
            return self.comm.world
        return dist.get_world_size(self.group) if (dist.is_available() and dist.is_initialized()) else 1

    def remove_hooks(self):
        if self._installed and getattr(self, "_tn_flags_before", None) is not None:
            from . import _lib
            if not (self._tn_flags_before & 2):          # clear only the bit this arena set; leave bits others changed meanwhile
                _lib.set_option("tn_flags", _lib.get_option("tn_flags") & ~2)
            self._tn_flags_before = None
        for first, _, _ in self.block_buckets:
            ops.grad_ready_hooks.pop(id(first), None)
        for h in self._hook_handles:
            h.remove()
        self._hook_handles.clear()
        if self._installed:
            ops.async_wgrad_blockers = max(0, ops.async_wgrad_blockers - 1)
        self._installed = False

    def _into_arena(self, p):
        """a gradient autograd produced outside the arena (an op without a fused write target) is moved into its slice"""
        if p.grad is not None and p.grad.data_ptr() != p._eoe_grad_buf.data_ptr():
            p._eoe_grad_buf.copy_(p.grad)
            p.grad = p._eoe_grad_buf

    def _arrived(self, p, b):
        self._into_arena(p)
        self._pending[b] -= 1
        if self._pending[b] == 0:
            ps, lo, hi = self.run_buckets[b]
            self._reduce_slice(lo, hi)

    def _reduce_slice(self, lo, hi):
        self.issued.append((lo, hi))
        if self.comm is None and self._world() <= 1:
            return
        buf = self.flat[lo:hi]
        if self.flat16 is not None:
            buf = self.flat16[lo:hi]
            buf.copy_(self.flat[lo:hi])                   # one cast pass on the compute stream, in front of the collective
            self._sent16.append((lo, hi))
        if self.comm is not None:
            self.comm.all_reduce_async(buf)
        else:
            self.handles.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        """all-reduce whatever was not reduced from inside backward, then wait for every collective of the step.  What is sent
        depends only on the model (never on which gradients happen to be None or outside the arena on this rank), so that all
        ranks issue the same sequence."""
        if self._installed:
            for b, (ps, lo, hi) in enumerate(self.run_buckets):
                if self._pending[b] != 0:                 # a run with a parameter that received no gradient this step
                    for p in ps:
                        self._into_arena(p)
                    self._reduce_slice(lo, hi)
                self._pending[b] = len(ps)
        else:
            for p in self.params:
                self._into_arena(p)
            self._reduce_slice(0, self.flat.numel())
        for h in self.handles:
            h.wait()
        self.handles.clear()
        if self.comm is not None:
            self.comm.join()
        for lo, hi in self._sent16:                       # the summed 16-bit buckets back into the fp32 arena the optimiser reads
            self.flat[lo:hi].copy_(self.flat16[lo:hi])
        self._sent16 = []
        self.issued = []


_bn_sync_cb = None            # the ctypes callback object must outlive its registration


class _DevPtr:
    """a raw device buffer as a `__cuda_array_interface__` object (what torch.as_tensor wraps without a copy)"""

    def __init__(self, ptr: int, count: int, f64: bool):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8" if f64 else "<f4", "data": (ptr, False), "version": 2}


def enable_sync_bn(process_group=None, comm: "NativeComm" = None) -> bool:
    """synchronised BatchNorm over `process_group` (module docstring); returns False (and clears the hook) when there is
    nothing to synchronise (no process group, or one rank).  With a `NativeComm` the sums are added by RCCL inside the library
    (`eoe_comm_sync_bn`: no Python in the path, same stream as the BatchNorm kernels)"""
    global _bn_sync_cb
    from . import _lib
    if comm is not None:
        # the BatchNorm sums get a second RCCL communicator; its id travels like the first one's (made on rank 0, broadcast by torch) -- once:
        # the library keeps the communicator, so a later enable draws no new id (no bootstrap thread, no broadcast)
        bn_id = None if getattr(comm, "_bn_ready", False) else comm.fresh_id()
        _lib.check(_lib.lib.eoe_comm_sync_bn(comm.handle, 1, bn_id), "eoe_comm_sync_bn")
        comm._bn_ready = True
        _bn_sync_cb = comm                    # keeps the communicator alive while registered
        return True
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(process_group) == 1:
        disable_sync_bn()
        return False

    def _hook(user, buf, count, is_f64, stream):
        try:
            # the library passes the stream the ops were given = torch's current stream, the one all_reduce orders itself after
            t = torch.as_tensor(_DevPtr(buf, count, bool(is_f64)), device=torch.device("cuda", torch.cuda.current_device()))
            dist.all_reduce(t, group=process_group)
            return 0
        except Exception as e:          # an exception must not unwind through the C frame
            print(f"eoe_amd.parallel: BatchNorm all-reduce failed: {e!r}", flush=True)
            return 1

    cb = _lib.ALLREDUCE_FN(_hook)
    _lib.check(_lib.lib.eoe_set_bn_sync(C.cast(cb, C.c_void_p), None), "eoe_set_bn_sync")
    _bn_sync_cb = cb
    return True


def disable_sync_bn():
    global _bn_sync_cb
    from . import _lib
    _lib.check(_lib.lib.eoe_set_bn_sync(None, None), "eoe_set_bn_sync")
    _bn_sync_cb = None


def all_gather_1d(t: torch.Tensor, group=None) -> torch.Tensor:
    """concatenate 1-D tensors from all ranks in rank order (scores / labels for the epoch AUC).  Lengths may differ: the
    reference keeps the ragged last batch (no drop_last, `bases.py:231-235`) and `shard_rows` splits it by floor, so ranks
    hold different numbers of rows (possibly none).  Lengths are exchanged first, payloads padded to the longest."""
    if not (dist.is_initialized() and dist.get_world_size(group) > 1):
        return t
    world = dist.get_world_size(group)
    t = t.contiguous().reshape(-1)
    n = torch.tensor([t.numel()], dtype=torch.int64, device=t.device)
    lens = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(lens, n, group=group)
    lens = [int(v.item()) for v in lens]
    cap = max(lens)
    if cap == 0:
        return t
    pad = torch.zeros(cap, dtype=t.dtype, device=t.device)
    pad[:t.numel()] = t
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad, group=group)
    return torch.cat([o[:k] for o, k in zip(out, lens)])
