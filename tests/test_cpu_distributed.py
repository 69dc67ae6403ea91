"""CPU tier: the N>1 path of the data-parallel layer with 2 gloo ranks: row sharding + loss/global-N scaling +
summed gradients through GradArena reproduce the single-process full-batch gradient; scores gather in rank order."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))


def _per_sample_loss(f, y):
    d = torch.sqrt((f * f).sum(1) + 1) - 1
    return torch.where(y == 0, d, -torch.log(1 - torch.exp(-d) + 1e-9))


def _worker(rank, world, port, out, hooks=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from eoe_amd import parallel
    r, w, _ = parallel.init_from_env("gloo")
    assert (r, w) == (rank, world)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(10, 6, generator=g)
    y = torch.cat([torch.zeros(5, dtype=torch.long), torch.ones(5, dtype=torch.long)])
    m = _model()
    arena = parallel.GradArena(m)
    if hooks:
        arena.install_hooks()          # the non-block parameters then go out as contiguous runs of the arena
    rows = parallel.shard_rows(5, 5, rank, world)
    loss = _per_sample_loss(m(x[rows]), y[rows]).sum() / 10.0          # sum(local) / GLOBAL batch
    loss.backward()
    # half of the parameters sit in the arena (as the fused kernels leave them), half are stray tensors
    for i, p in enumerate(m.parameters()):
        if i % 2 == 0:
            p._eoe_grad_buf.copy_(p.grad)
            p.grad = p._eoe_grad_buf
    arena.finish()
    scores = parallel.all_gather_1d(torch.full((3,), float(rank)))
    tot = torch.tensor([loss.item()])
    dist.all_reduce(tot)
    if rank == 0:
        torch.save({"grads": [p.grad.clone() for p in m.parameters()], "scores": scores, "loss": tot}, out)
    dist.barrier()
    dist.destroy_process_group()


import pytest  # noqa: E402


@pytest.mark.parametrize("hooks", [False, True])
def test_two_rank_gradients_equal_full_batch(tmp_path, hooks):
    out = str(tmp_path / "r0.pt")
    port = _free_port()
    mp.spawn(_worker, args=(2, port, out, hooks), nprocs=2, join=True)
    got = torch.load(out)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(10, 6, generator=g)
    y = torch.cat([torch.zeros(5, dtype=torch.long), torch.ones(5, dtype=torch.long)])
    m = _model()
    loss = _per_sample_loss(m(x), y).mean()
    loss.backward()
    for a, p in zip(got["grads"], m.parameters()):
        np.testing.assert_allclose(a.numpy(), p.grad.numpy(), rtol=1e-5, atol=1e-7)
    assert abs(got["loss"].item() - loss.item()) < 1e-6
    assert got["scores"].tolist() == [0.0] * 3 + [1.0] * 3


def _ragged_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from eoe_amd import parallel
    parallel.init_from_env("gloo")
    # ranks hold 3 / 0 / 2 rows (a ragged last batch split by floor; one rank has nothing)
    mine = {0: torch.tensor([0.0, 1.0, 2.0]), 1: torch.zeros(0), 2: torch.tensor([7.0, 8.0])}[rank]
    got = parallel.all_gather_1d(mine)
    lab = parallel.all_gather_1d(torch.arange(len(mine), dtype=torch.int64))
    if rank == 0:
        torch.save({"vals": got, "labels": lab}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_all_gather_1d_ragged_lengths(tmp_path):
    out = str(tmp_path / "rag.pt")
    mp.spawn(_ragged_worker, args=(3, _free_port(), out), nprocs=3, join=True)
    got = torch.load(out)
    assert got["vals"].tolist() == [0.0, 1.0, 2.0, 7.0, 8.0]
    assert got["labels"].tolist() == [0, 1, 2, 0, 1]


def _bucket_worker(rank, world, port, out):
    """a model without fused blocks: every parameter goes out in `post_accumulate_grad` run buckets; one parameter receives no
    gradient on either rank (its run is still sent, so that both ranks issue the same collectives)"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from eoe_amd import parallel
    parallel.init_from_env("gloo")
    torch.manual_seed(0)
    m = torch.nn.Sequential(torch.nn.Linear(6, 32), torch.nn.Tanh(), torch.nn.Linear(32, 32), torch.nn.Tanh(), torch.nn.Linear(32, 3))
    unused = torch.nn.Parameter(torch.zeros(40))
    m.register_parameter("unused", unused)
    arena = parallel.GradArena(m, bucket_bytes=1024)
    arena.install_hooks()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(10, 6, generator=g)
    rows = parallel.shard_rows(5, 5, rank, world)
    (m(x[rows]).pow(2).sum() / 10.0).backward()
    issued = list(arena.issued)
    arena.finish()
    # second step: the buckets re-arm
    for p in m.parameters():
        p.grad = None
    (m(x[rows]).pow(2).sum() / 10.0).backward()
    issued2 = list(arena.issued)
    arena.finish()
    if rank == 0:
        torch.save({"grads": [None if p.grad is None else p.grad.clone() for p in m.parameters()], "issued": issued,
                    "issued2": issued2, "buckets": [(lo, hi) for _, lo, hi in arena.run_buckets]}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_run_buckets_issue_from_backward(tmp_path):
    out = str(tmp_path / "bk.pt")
    mp.spawn(_bucket_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    assert len(got["buckets"]) >= 3
    # all buckets but the one holding the gradient-less parameter were sent from inside backward, last layer first; both steps alike
    assert len(got["issued"]) == len(got["buckets"]) - 1 and got["issued"] == got["issued2"]
    assert got["issued"][0][0] > got["issued"][-1][0]
    torch.manual_seed(0)
    m = torch.nn.Sequential(torch.nn.Linear(6, 32), torch.nn.Tanh(), torch.nn.Linear(32, 32), torch.nn.Tanh(), torch.nn.Linear(32, 3))
    g = torch.Generator().manual_seed(1)
    x = torch.randn(10, 6, generator=g)
    (m(x).pow(2).sum() / 10.0).backward()
    assert got["grads"][0] is None                      # `unused` registers on the container itself: first in parameters()
    for a, p in zip(got["grads"][1:], m.parameters()):
        np.testing.assert_allclose(a.numpy(), p.grad.numpy(), rtol=1e-5, atol=1e-7)
