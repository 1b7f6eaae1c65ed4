"""How a rank draws its shard of the training set: the semantics of the reference's data pipeline around the hot path
(main.py:79 `DistributedSampler(train_dataset)`, main.py:80-88 `DataLoader(..., batch_size, sampler, drop_last=True)`,
main.py:108 `train_sampler.set_epoch(epoch)`), without the DataLoader worker machinery.

  * shuffle with a generator seeded `seed + epoch` -- every rank draws the SAME permutation;
  * pad the permutation to a multiple of the world size by repeating its head (sampler `drop_last=False`, the reference's
    default), or cut it down to one (sampler `drop_last=True`);
  * rank r takes positions r, r + world, r + 2 world, ...;
  * batches of `batch_size` consecutive picks; the last partial batch of a rank is dropped (loader `drop_last=True`).

`distributed_shard` reproduces `torch.utils.data.DistributedSampler.__iter__` index for index (tests/test_host_cpu.py);
`ShardedLoader` is what `Trainer.train(epoch, loader, device)` iterates: it calls `set_epoch(epoch)` itself, like main.py:108.
"""
import math

import torch


def distributed_shard(n, rank, world, epoch=0, seed=0, shuffle=True, drop_last=False):
    """-> list of dataset indices rank `rank` of `world` visits in epoch `epoch` (DistributedSampler semantics)."""
    if not 0 <= rank < world:
        raise ValueError("distributed_shard: rank %d outside [0, %d)" % (rank, world))
    if shuffle:
        g = torch.Generator()
        g.manual_seed(seed + epoch)
        idx = torch.randperm(n, generator=g).tolist()
    else:
        idx = list(range(n))
    if drop_last and n % world != 0:
        per = math.ceil((n - world) / world)
    else:
        per = math.ceil(n / world)
    total = per * world
    if not drop_last:
        pad = total - len(idx)
        if pad > 0:
            idx += (idx * math.ceil(pad / max(len(idx), 1)))[:pad]
    else:
        idx = idx[:total]
    return idx[rank:total:world]


def _collate(samples):
    """Stack a list of per-sample dicts (tensors stacked, everything else listed) -- the default collate for dict samples."""
    out = {}
    for k in samples[0]:
        v = [s[k] for s in samples]
        out[k] = torch.stack(v) if torch.is_tensor(v[0]) else v
    return out


class ShardedLoader:
    """Iterates this rank's batches of an indexable dataset (`dataset[i]` -> dict of tensors, or a dict of batched tensors
    whose first axis is the sample axis).  `set_epoch` reshuffles (main.py:108)."""

    def __init__(self, dataset, batch_size, rank=0, world=1, seed=0, shuffle=True, sampler_drop_last=False, drop_last=True,
                 collate=_collate):
        self.dataset, self.batch_size = dataset, int(batch_size)
        self.rank, self.world, self.seed, self.shuffle = rank, world, seed, shuffle
        self.sampler_drop_last, self.drop_last, self.collate = sampler_drop_last, drop_last, collate
        self.epoch = 0
        self._columns = isinstance(dataset, dict)
        self.n = len(next(iter(dataset.values()))) if self._columns else len(dataset)

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def indices(self):
        return distributed_shard(self.n, self.rank, self.world, self.epoch, self.seed, self.shuffle, self.sampler_drop_last)

    def __len__(self):
        m = len(self.indices())
        return m // self.batch_size if self.drop_last else math.ceil(m / self.batch_size)

    def __iter__(self):
        idx = self.indices()
        B = self.batch_size
        stop = len(idx) - (len(idx) % B if self.drop_last else 0)
        for i in range(0, stop, B):
            pick = idx[i:i + B]
            if self._columns:
                sel = torch.as_tensor(pick)
                yield {k: (v[sel.to(v.device)] if torch.is_tensor(v) else [v[j] for j in pick]) for k, v in self.dataset.items()}
            else:
                yield self.collate([self.dataset[j] for j in pick])
