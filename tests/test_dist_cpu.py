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
        w.finish()
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


def _reducer_worker(rank, world, port, q, comm):
    """The trainer's own reduction object (GradReducer: what Trainer._early_grads_ready / train_step drive) on a fake
    backward: the early slice is complete when the hook fires, the late slice is written afterwards, the never-used tail
    never; with and without bf16 transport."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pdfnet_amd.trains.base_trainer import GradReducer
    torch.manual_seed(7 + rank)
    n, n_early, n_live = 50000, 30016, 44032
    flat = torch.zeros(n)
    red = GradReducer(flat, n_early, n_live, comm_dtype=torch.bfloat16 if comm == 'bf16' else None)
    errs = []
    for step in range(2):                                    # the reducer re-arms every step
        red.reset()
        flat.zero_()
        mine = torch.zeros(n)
        mine[:n_live] = torch.randn(n_live)
        mine[n_live:] = 0.0                                  # parameters no autograd path reaches
        flat[:n_early] = mine[:n_early]                      # ... backward of everything above the trunk
        red.early_ready()                                    # hook on the trunk output's gradient
        red.early_ready()                                    # (firing twice must not send twice)
        flat[n_early:n_live] = mine[n_early:n_live]          # ... trunk backward, while the early slice is in flight
        red.finish()
        gathered = [torch.empty(n) for _ in range(world)]
        dist.all_gather(gathered, mine)
        ref = sum(gathered)
        errs.append(float((flat - ref).abs().max() / ref.abs().max()))
        assert float(flat[n_live:].abs().max()) == 0.0
    q.put((rank, max(errs), red.bytes_sent))
    dist.destroy_process_group()


def _run_reducer(comm):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_reducer_worker, args=(r, 2, port, q, comm)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    return res


def test_trainer_reduction_object_world2_fp32():
    res = _run_reducer('fp32')
    assert all(e < 1e-6 for _, e, _ in res), res
    assert all(b == 44032 * 4 for _, _, b in res)            # the never-used tail is not sent


def test_trainer_reduction_object_world2_bf16_transport():
    res = _run_reducer('bf16')
    assert all(e < 1e-2 for _, e, _ in res), res             # bf16 rounding of the summands: 2^-8 relative
    assert all(b == 44032 * 2 for _, _, b in res)


def _shard_worker(rank, world, port, q):
    """Two ranks draw their shard of one epoch the way Trainer.train does (ShardedLoader.set_epoch, main.py:79,88,108)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pdfnet_amd.trains.sampler import ShardedLoader
    n, B = 37, 4                                             # odd dataset: the permutation is padded to 38 by repeating its head
    data = {'idx': torch.arange(n), 'x': torch.arange(n, dtype=torch.float32) * 2}
    ld = ShardedLoader(data, batch_size=B, rank=dist.get_rank(), world=dist.get_world_size(), seed=0)
    per_epoch = []
    for epoch in (1, 2):
        ld.set_epoch(epoch)
        picks = torch.cat([b['idx'] for b in ld])
        assert len(ld) == 19 // B and picks.numel() == (19 // B) * B          # the rank's 19 picks -> 4 full batches, the rest dropped
        every = torch.tensor(ld.indices())
        gathered = [torch.empty_like(every) for _ in range(world)]
        dist.all_gather(gathered, every)
        per_epoch.append((picks.tolist(), [g.tolist() for g in gathered]))
    q.put((rank, per_epoch))
    dist.destroy_process_group()


def test_two_ranks_draw_disjoint_padded_shards_that_change_with_the_epoch():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_shard_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    for e in range(2):
        a, b = res[0][e][1]                                  # both ranks' full index streams, as rank 0 gathered them
        assert res[1][e][1] == [a, b]                        # ... and rank 1 saw the same
        assert len(a) == len(b) == 19
        union = a + b
        assert set(union) == set(range(37)) and len(union) == 38          # every sample once, one repeated (the padding)
        assert len(set(a) & set(b)) <= 1                     # disjoint up to the padded element
        assert res[0][e][0] == a[:16] and res[1][e][0] == b[:16]           # batches = the shard in order, last partial batch dropped
    assert res[0][0][1] != res[0][1][1]                      # set_epoch reshuffles
