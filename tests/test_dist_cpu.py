"""CPU (gloo, world_size 2): the N>1 gradient path of the trainer -- chunked SUM all-reduces over the early and the late
slice of the flat gradient buffer followed by 1/world scaling -- gives every rank the mean of the per-rank gradients."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pdfnet_amd.trains.base_trainer import allreduce_flat_grads
    torch.manual_seed(100 + rank)
    n = 100003                                   # not divisible by the chunk count
    g = torch.randn(n)
    mine = g.clone()
    # the trainer's overlapped form: the early slice goes out first without waiting (from inside the backward), the late
    # slice follows after the backward, then both are waited for
    n_early = 66001
    early = allreduce_flat_grads(g[:n_early], chunks=3, wait=False)
    assert len(early) == 3
    assert allreduce_flat_grads(g[n_early:], chunks=2) == []
    for w in early:
        w.wait()
    g *= 1.0 / world                             # FlatAdam's grad_scale
    gathered = [torch.empty(n) for _ in range(world)]
    dist.all_gather(gathered, mine)
    ref = sum(gathered) / world
    q.put((rank, float((g - ref).abs().max())))
    dist.destroy_process_group()


def test_flat_gradient_allreduce_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    assert sorted(r for r, _ in res) == [0, 1]
    assert all(e < 1e-6 for _, e in res), res


def test_single_process_is_a_noop():
    from pdfnet_amd.trains.base_trainer import allreduce_flat_grads
    g = torch.arange(10.0)
    allreduce_flat_grads(g)
    assert torch.equal(g, torch.arange(10.0))
