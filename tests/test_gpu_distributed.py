"""GPU tier: the data-parallel path end to end with TWO ranks sharing the one GPU of the test box (gloo backend on
CUDA tensors, since RCCL refuses two ranks on one device): per-block gradient buckets all-reduced from inside
backward + sum/global-N loss scaling reproduce the single-process full-batch gradients of the fused ViT."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make(seed=0):
    import eoe_amd
    from eoe_amd.models import ClipViTB32Custom
    from oracle import models as omodels
    eoe_amd.set_compute_dtype("fp16")
    m = omodels.deterministic_init(ClipViTB32Custom(layers=2), tag="ddp", layers=2).cuda().train()
    return m


def _batch():
    from oracle import trainer as otrainer
    return otrainer.synthetic_batch("ddp/b", 4, 4, 224)


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch.distributed as dist
    import eoe_amd
    from eoe_amd import parallel
    parallel.init_from_env("gloo")
    torch.cuda.set_device(0)
    m = _make()
    arena = parallel.GradArena(m)
    arena.install_hooks()
    x, y = _batch()
    rows = parallel.shard_rows(4, 4, rank, world)
    feats = m(x[rows].cuda())
    loss = eoe_amd.hsc_loss(feats, y[rows].cuda(), 0, 1.0 / 8)
    loss.backward()
    arena.finish()
    in_arena = sum(1 for p in m.parameters() if p.grad.data_ptr() == p._eoe_grad_buf.data_ptr())
    tot = loss.detach().clone()
    dist.all_reduce(tot)
    if rank == 0:
        torch.save({"grads": {k: p.grad.cpu() for k, p in m.named_parameters()}, "loss": tot.cpu(), "in_arena": in_arena,
                    "n_params": len(list(m.parameters()))}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_match_single_process(tmp_path):
    import eoe_amd
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    m = _make()
    x, y = _batch()
    loss = eoe_amd.hsc_loss(m(x.cuda()), y.cuda(), 0)
    loss.backward()
    assert abs(got["loss"].item() - loss.item()) < 1e-4 * max(1.0, abs(loss.item()))
    # gradients written by the fused kernels must have been adopted in place (no copy) for the bucketed all-reduce
    assert got["in_arena"] == got["n_params"], (got["in_arena"], got["n_params"])
    for k, p in m.named_parameters():
        ref = p.grad.cpu().double()
        err = (got["grads"][k].double() - ref).norm().item() / max(ref.norm().item(), 1e-12)
        assert err < 5e-3, (k, err)


# ------------------------------------------------------------------------------------------------ RCCL itself (one rank: the box has one GPU)
def _rccl_worker(rank, world, port, out):
    """world_size 1 over the REAL RCCL: (a) torch.distributed backend "nccl" driving GradArena's buckets (ViT block hooks +
    generic run buckets) and the ragged all_gather_1d; (b) the C-ABI communicator (eoe_comm_*, csrc/comm.cpp) driving the same
    arena, ring and reduce-scatter + all-gather forms, plus its all-gather.  With one rank a SUM all-reduce must return the
    buffer unchanged -- what is exercised is that the library loads RCCL, builds communicators, and that the stream / event
    ordering around the collectives is right (results are read right after `finish`)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    import eoe_amd
    from eoe_amd import parallel, _lib
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    res = {}
    x, y = _batch()
    for mode in ("torch_nccl", "native_ring", "native_rs_ag"):
        m = _make()
        comm = None if mode == "torch_nccl" else parallel.NativeComm(
            algo=_lib.EOE_COMM_ALGO_RS_AG if mode == "native_rs_ag" else _lib.EOE_COMM_ALGO_RING)
        arena = parallel.GradArena(m, comm=comm)
        arena.install_hooks()
        # force real collectives even at world 1 on the torch path
        if comm is None:
            arena._reduce_slice = lambda lo, hi, a=arena: (a.issued.append((lo, hi)),
                                                           a.handles.append(dist.all_reduce(a.flat[lo:hi], async_op=True)))[0]
        loss = eoe_amd.hsc_loss(m(x.cuda()), y.cuda(), 0)
        loss.backward()
        n_coll = len(arena.issued)
        arena.finish()
        res[mode] = {"grads": {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters()}, "collectives": n_coll + 0}
        arena.remove_hooks()
        if comm is not None:
            g = comm.all_gather_async(torch.arange(5, device="cuda", dtype=torch.float32))
            comm.join()
            torch.cuda.synchronize()
            res[mode]["gather"] = g.cpu()
            comm.close()
    res["ragged"] = parallel.all_gather_1d(torch.arange(3, device="cuda", dtype=torch.float32)).cpu()
    # (c) synchronised BatchNorm with RCCL called from inside the library (eoe_comm_sync_bn) and through the torch.distributed hook
    # (world 1 is "nothing to synchronise" for the latter: forced on by binding the hook by hand)
    from eoe_amd.models import CNN32
    from oracle import models as omodels, trainer as otrainer
    xb, yb = otrainer.synthetic_batch("ddp/syncbn1", 8, 8, 32)
    for mode in ("plain", "native"):
        m = omodels.deterministic_init(CNN32(bias=True), tag="cnn32").cuda().train()
        comm = parallel.NativeComm() if mode == "native" else None
        if comm is not None:
            assert parallel.enable_sync_bn(comm=comm)
        eoe_amd.hsc_loss(m(xb.cuda()), yb.cuda(), 0).backward()
        torch.cuda.synchronize()
        res["bn_" + mode] = {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters()}
        res["bn_" + mode].update({k: v.detach().cpu().clone() for k, v in m.state_dict().items() if "running" in k})
        if comm is not None:
            parallel.disable_sync_bn()
            comm.close()
    torch.save(res, out)
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_world1_buckets_and_native_comm(tmp_path):
    import eoe_amd
    out = str(tmp_path / "rccl.pt")
    mp.spawn(_rccl_worker, args=(1, _free_port(), out), nprocs=1, join=True)
    got = torch.load(out)
    m = _make()
    x, y = _batch()
    eoe_amd.hsc_loss(m(x.cuda()), y.cuda(), 0).backward()
    for mode in ("torch_nccl", "native_ring", "native_rs_ag"):
        assert got[mode]["collectives"] >= 3, got[mode]["collectives"]           # 2 block buckets + at least one run bucket
        for k, p in m.named_parameters():
            assert torch.equal(got[mode]["grads"][k], p.grad.cpu()), (mode, k)   # the step is bitwise reproducible
    assert got["native_ring"]["gather"].tolist() == [[0.0, 1.0, 2.0, 3.0, 4.0]]
    assert got["ragged"].tolist() == [0.0, 1.0, 2.0]
    # one rank: the all-reduced BatchNorm sums are the local ones (forward sums pass through double precision on the way)
    for k, v in got["bn_plain"].items():
        a, b = got["bn_native"][k].double(), v.double()
        assert (a - b).norm().item() <= 1e-5 * max(b.norm().item(), 1e-6), k


def _cnn_worker(rank, world, port, out):
    """2 gloo ranks on the one GPU: CNN32's layers go out as `post_accumulate_grad` run buckets while backward continues"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch.distributed as dist
    import eoe_amd
    from eoe_amd import parallel
    from eoe_amd.models import CNN32
    from oracle import models as omodels, trainer as otrainer
    parallel.init_from_env("gloo")
    torch.cuda.set_device(0)
    m = omodels.deterministic_init(CNN32(bias=True), tag="cnn32").cuda().train()
    arena = parallel.GradArena(m, bucket_bytes=256 << 10)
    arena.install_hooks()
    x, y = otrainer.synthetic_batch("ddp/cnn", 16, 16, 32)
    # every rank computes the FULL batch weighted 1/world (what the trainer does for batches too small to shard): BatchNorm
    # statistics are then the full-batch ones and the summed gradient must equal the single-process gradient
    loss = eoe_amd.hsc_loss(m(x.cuda()), y.cuda(), 0, 1.0 / (32 * world))
    loss.backward()
    issued = list(arena.issued)
    arena.finish()
    if rank == 0:
        torch.save({"grads": {k: p.grad.cpu() for k, p in m.named_parameters()}, "issued": issued,
                    "n_buckets": len(arena.run_buckets)}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_cnn32_run_buckets_two_ranks(tmp_path):
    import eoe_amd
    from eoe_amd.models import CNN32
    from oracle import models as omodels, trainer as otrainer
    out = str(tmp_path / "cnn.pt")
    mp.spawn(_cnn_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    assert got["n_buckets"] >= 3 and len(got["issued"]) == got["n_buckets"]      # every bucket went out from inside backward
    assert got["issued"][0][0] > got["issued"][-1][0]                            # deep layers first
    m = omodels.deterministic_init(CNN32(bias=True), tag="cnn32").cuda().train()
    x, y = otrainer.synthetic_batch("ddp/cnn", 16, 16, 32)
    eoe_amd.hsc_loss(m(x.cuda()), y.cuda(), 0).backward()
    for k, p in m.named_parameters():
        ref = p.grad.cpu().double()
        err = (got["grads"][k].double() - ref).norm().item() / max(ref.norm().item(), 1e-9)
        # each rank's loss is scaled by 1 / world, which moves the fp16 rounding of the dY operands (4e-4 measured)
        # (biases in front of a BatchNorm have a true gradient of 0: both sides hold rounding noise only)
        noise_only = k in ("conv1.bias", "conv2.bias", "conv3.bias", "fc1.bias")
        assert err < 3e-3 or noise_only, (k, err, ref.norm().item())


# ------------------------------------------------------------------------------------------------ the trainer itself under data parallelism
def _trainer_batches():
    from oracle import trainer as otrainer
    # full batch (4 + 4), a ragged one whose halves split unevenly over 2 ranks (3 + 3), and one smaller than the ranks (1 + 1)
    return [otrainer.synthetic_batch("ddp/t0", 4, 4, 224), otrainer.synthetic_batch("ddp/t1", 3, 3, 224),
            otrainer.synthetic_batch("ddp/t2", 1, 1, 224)]


def _trainer_run(data_parallel):
    import eoe_amd
    from eoe_amd.data import ListSource
    from eoe_amd.training import HSCTrainer
    eoe_amd.set_compute_dtype("fp16")
    m = _make()
    tr = HSCTrainer(m, dataset=ListSource(_trainer_batches()), epochs=2, lr=1e-4, wdk=1e-3, batch_size=4, data_parallel=data_parallel)
    model = tr._fresh_model(m)            # a module preset: used as is (no weight reset), like `load` does
    model, roc = tr.train_cls(model, tr.ds, 0, "0", 0)
    return tr.last_losses, roc.auc, {k: v.detach().cpu() for k, v in model.state_dict().items()}


def _trainer_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch.distributed as dist
    from eoe_amd import parallel
    parallel.init_from_env("gloo")
    torch.cuda.set_device(0)
    losses, auc, sd = _trainer_run(True)
    if rank == 0:
        torch.save({"losses": losses, "auc": auc, "sd": sd}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_trainer_data_parallel_ragged_batches(tmp_path):
    """`train_cls(data_parallel=True)` with 2 ranks over an epoch whose batches are full (4 + 4), uneven over the ranks (3 + 3: one
    rank gets 1 + 1 rows, the other 2 + 2) and smaller than the world in each half (1 + 1: computed whole on every rank, weighted
    1 / world): the per-step losses, the epoch AUC and the trained weights equal the single-process run (the reference keeps the
    ragged last batch: no drop_last, bases.py:231-235)"""
    out = str(tmp_path / "tr.pt")
    mp.spawn(_trainer_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    losses, auc, sd = _trainer_run(False)
    np.testing.assert_allclose(got["losses"], losses, rtol=2e-3, atol=1e-5)
    assert abs(got["auc"] - auc) < 1e-6
    for k in sd:
        a, b = got["sd"][k].double(), sd[k].double()
        assert (a - b).norm().item() <= 2e-3 * max(b.norm().item(), 1e-6), k


# ------------------------------------------------------------------------------------------------ synchronised BatchNorm
def _bn_model(kind):
    from eoe_amd.models import CNN32, WideResNet
    from oracle import models as omodels
    if kind == "cnn32":
        return omodels.deterministic_init(CNN32(bias=True), tag="cnn32").cuda().train(), (15, 15, 32)
    return omodels.deterministic_init(WideResNet(res=32), tag="wrn32sync").cuda().train(), (7, 7, 32)


def _bn_buffers(m):
    return {k: v.detach().cpu() for k, v in m.state_dict().items() if "running_" in k or "num_batches" in k}


def _syncbn_worker(rank, world, port, out, kind):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch.distributed as dist
    import eoe_amd
    from eoe_amd import parallel
    from oracle import trainer as otrainer
    parallel.init_from_env("gloo")
    torch.cuda.set_device(0)
    eoe_amd.set_compute_dtype("fp16")
    eoe_amd.set_parity_mode(kind == "wrn32")
    m, (nn_, no_, res) = _bn_model(kind)
    arena = parallel.GradArena(m, bucket_bytes=256 << 10)
    arena.install_hooks()
    assert parallel.enable_sync_bn()
    x, y = otrainer.synthetic_batch("ddp/syncbn", nn_, no_, res)
    rows = parallel.shard_rows(nn_, no_, rank, world)              # odd halves: the ranks hold 7 + 7 and 8 + 8 rows (3 + 3 / 4 + 4)
    loss = eoe_amd.hsc_loss(m(x[rows].cuda()), y[rows].cuda(), 0, 1.0 / (nn_ + no_))
    loss.backward()
    arena.finish()
    parallel.disable_sync_bn()
    tot = loss.detach().clone()
    dist.all_reduce(tot)
    if rank == 0:
        torch.save({"grads": {k: p.grad.cpu() for k, p in m.named_parameters()}, "loss": tot.cpu(), "buffers": _bn_buffers(m),
                    "rows": len(rows)}, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["cnn32", "wrn32"])
def test_sync_batchnorm_two_ranks_equal_the_full_batch(tmp_path, kind):
    """2 ranks with UNEVEN shards and `enable_sync_bn()`: every BatchNorm (2-d, 1-d, the spatial gate's one-channel one) normalises
    with the statistics of the global batch, so loss, summed gradients and the running buffers equal the single-process full-batch
    step -- the meaning the single-device reference gives its BatchNorm layers (cnn.py:57-66, resnet.py:37-41, cbam.py:74)"""
    import eoe_amd
    from oracle import trainer as otrainer
    out = str(tmp_path / "sbn.pt")
    mp.spawn(_syncbn_worker, args=(2, _free_port(), out, kind), nprocs=2, join=True)
    got = torch.load(out)
    eoe_amd.set_compute_dtype("fp16")
    # the 14-image WideResNet (1x1 maps in layer4: BatchNorm over 14 values) amplifies 16-bit operand rounding to percents of the
    # stem gradient whichever way the batch is split; its fp32 parity mode isolates the BatchNorm logic under test
    eoe_amd.set_parity_mode(kind == "wrn32")
    try:
        m, (nn_, no_, res) = _bn_model(kind)
        assert got["rows"] < nn_ + no_
        x, y = otrainer.synthetic_batch("ddp/syncbn", nn_, no_, res)
        loss = eoe_amd.hsc_loss(m(x.cuda()), y.cuda(), 0)
        loss.backward()
    finally:
        eoe_amd.set_parity_mode(False)
    assert abs(got["loss"].item() - loss.item()) < 1e-4 * max(1.0, abs(loss.item()))
    for k, v in _bn_buffers(m).items():
        # (deep layers: the 16-bit activation copies upstream may round differently when a statistic moves by one ulp)
        np.testing.assert_allclose(got["buffers"][k].double().numpy(), v.double().numpy(), rtol=1e-3, atol=2e-5, err_msg=k)
    worst = 0.0
    for k, p in m.named_parameters():
        ref = p.grad.cpu().double()
        err = (got["grads"][k].double() - ref).norm().item() / max(ref.norm().item(), 1e-9)
        # (a bias in front of a BatchNorm has a true gradient of 0: both sides hold rounding noise only)
        noise_only = ref.norm().item() < 1e-6 * max(p.detach().norm().item(), 1.0) or (k.endswith("bias") and ("conv" in k or k == "fc1.bias") and kind == "cnn32")
        if not noise_only:
            worst = max(worst, err)
            assert err < (3e-3 if kind == "cnn32" else 1e-3), (k, err, ref.norm().item())
    print(f"[sync bn {kind}] worst relative gradient deviation {worst:.2e}")


def _run_bench(args, env_extra=None, nproc=1, timeout=900):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.update(env_extra or {})
    if nproc > 1:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.join(root, "bench.py")] + args
    else:
        cmd = [sys.executable, os.path.join(root, "bench.py")] + args
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_itself_with_two_ranks():
    """bench.py as the driver launches it for N > 1 (torchrun, one rank per process), rehearsed with two gloo ranks on the one GPU:
    ONE JSON line from rank 0, whole-job throughput, the exchange step described (transport, buckets, bytes, exposed wait)"""
    out = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--layers", "1", "--batch", "4", "--no-roofline"],
                     {"EOE_DIST_BACKEND": "gloo"}, nproc=2)
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["config"]["global_batch"] == 16
    c = out["comm"]
    assert c["transport"].startswith("torch.distributed (gloo") and c["buckets"] >= 2 and c["comm_exposed_ms"] >= 0.0
    # ViT-B/32 with one block: every trainable parameter is sent exactly once (slices are padded to 256 B)
    from eoe_amd.models import ClipViTB32Custom
    n_par = sum(p.numel() for p in ClipViTB32Custom(layers=1).parameters())
    assert 4 * n_par <= c["allreduce_bytes_per_step"] <= 4 * n_par * 1.01 + 65536
    assert out["value"] > 0 and np.isfinite(out["final_loss"])


def test_bench_eval_mode_line():
    """`bench.py --mode eval`: the forward-only scoring loop of eval_cls (ad_trainer.py:473-550) as a throughput line"""
    out = _run_bench(["--mode", "eval", "--steps", "3", "--warmup", "1", "--layers", "2", "--batch", "8", "--no-cpu-baseline"])
    assert out["metric"].startswith("eval images/sec") and out["value"] > 0 and out["final_loss"] is None
    assert "eval_cls" in out["config"]["workload"]
    r = out["roofline"]
    assert r["kernel"].startswith("gemm") and set(k for k in r["kernels_ms_per_step"] if k.startswith("gemm")) == {"gemm_nt"}   # no wgrad
