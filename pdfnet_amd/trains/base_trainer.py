"""Counterpart of the reference's train loop (lib/trains/base_trainer.py:24-199, main.py:63-148): model+loss
wrapper, Adam(lr=1e-4) over model parameters, one process per GPU with a gradient all-reduce over RCCL.

MI355X-first differences from the reference loop:
  * parameters, gradients and Adam moments live in three flat fp32 buffers (views keep every tensor's layout),
    so the optimizer is ONE fused kernel and the gradient all-reduce is a few large RCCL calls on a
    contiguous buffer instead of DDP's 25 MB buckets over 643 tensors;
  * the 324 parameters that never receive a gradient (reference needs find_unused_parameters=True,
    base_trainer.py:95) simply keep a zero gradient in the flat buffer -- no per-iteration graph walk;
  * the step has no host synchronisation, so forward+loss+backward (and the optimizer) replay as a hipGraph.
"""
import os

import torch
import torch.distributed as dist

from .. import functional as F
from .. import hip


class ModleWithLoss(torch.nn.Module):
    """base_trainer.py:24-78 (same class name): train -> (loss[B], stats); val/test -> the loss module's 9-tuple."""

    def __init__(self, model, loss):
        super().__init__()
        self.model = model
        self.loss = loss

    def forward(self, batch, mode, epoch):
        ind = batch['ind'] if mode == 'train' else None
        result, paramsDict, handDictList, otherInfo = self.model(
            batch['input'], batch['choose'], batch['cloud'], batch.get('depth'), ind, batch['K_new'], batch['valid'])
        return self.loss(result, paramsDict, handDictList, otherInfo, batch, mode, epoch)


class FlatAdam:
    """torch.optim.Adam(params, lr) semantics (main.py:63) on flat buffers with one fused HIP kernel."""

    def __init__(self, params, lr=1e-4, betas=(0.9, 0.999), eps=1e-8):
        self.params = [p for p in params]
        dev = self.params[0].device
        assert dev.type == 'cuda', "FlatAdam: parameters must be on the GPU (no CPU fallback)"
        self.lr, self.betas, self.eps = lr, betas, eps
        ALIGN = 64                                     # every tensor starts on a 256-byte boundary: the GEMM fast
        offs, n = [], 0                                # path needs 16-byte aligned operands (float4 loads)
        for p in self.params:
            offs.append(n)
            n += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        self.numel = n
        self.offsets = offs                            # flat offset of params[i] (the buffers follow the order given)
        self.flat_p = torch.zeros(n, device=dev)
        self.flat_g = torch.zeros(n, device=dev)
        self.flat_m = torch.zeros(n, device=dev)
        self.flat_v = torch.zeros(n, device=dev)
        for p, o in zip(self.params, offs):
            k = p.numel()
            view = self.flat_p[o:o + k].as_strided(p.shape, p.stride())      # keeps e.g. channels_last storage
            view.copy_(p.data)
            p.data = view
            p.grad = self.flat_g[o:o + k].as_strided(p.shape, p.stride())
            p._pdf_main_grad = True                   # HIP backward kernels accumulate straight into this view
        self.step_t = torch.zeros(1, device=dev)                             # device-side step count (graph-safe)
        self.corr = torch.ones(2, device=dev)
        self._b = torch.tensor(betas, device=dev)

    def zero_grad(self):
        self.flat_g.zero_()

    def step(self, grad_scale=1.0):
        F.join_wgrad()                                 # side-stream gradient kernels must have landed (no-op when already joined)
        self.step_t += 1
        torch.sub(1.0, torch.pow(self._b, self.step_t), out=self.corr)        # [1-b1^t, 1-b2^t]
        hip.lib().pdf_adam_step(hip.ptr(self.flat_p), hip.ptr(self.flat_g), hip.ptr(self.flat_m), hip.ptr(self.flat_v),
                                self.numel, self.lr, self.betas[0], self.betas[1], self.eps, hip.ptr(self.corr),
                                float(grad_scale), hip.stream())

    def state_dict(self):
        return {'step': self.step_t.clone(), 'exp_avg': self.flat_m.clone(), 'exp_avg_sq': self.flat_v.clone(), 'lr': self.lr}

    def load_state_dict(self, sd):
        self.step_t.copy_(sd['step'])
        self.flat_m.copy_(sd['exp_avg'])
        self.flat_v.copy_(sd['exp_avg_sq'])
        self.lr = sd.get('lr', self.lr)


def allreduce_flat_grads(flat_g, chunks=4, wait=True, force=False):
    """Gradient SUM across ranks (the mean's 1/world is folded into the optimizer's grad_scale).  A few large
    RCCL all-reduces on the contiguous buffer: xGMI rings are per-link bound, so fewer/larger beats many/small.
    wait=False returns the outstanding work handles instead of waiting for them; force=True issues the collectives on
    a one-rank group too (tests: exercises the RCCL calls on a one-GPU box)."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force) or flat_g.numel() == 0:
        return []
    n = flat_g.numel()
    per = (n + chunks - 1) // chunks
    works = [dist.all_reduce(flat_g[i:min(n, i + per)], op=dist.ReduceOp.SUM, async_op=True) for i in range(0, n, per)]
    if not wait:
        return works
    for w in works:
        w.wait()
    return []


# Parameters whose gradients are complete only at the very end of the backward: the ResNet trunk and what hangs off the
# stem (e_conv1, and PointNet++ levels 1-2, whose backward runs last on its side stream).  Everything else -- two thirds
# of the buffer: pyramid laterals, feat, heads, dense decoders, centre convolutions, SFT, mesh decoder -- is complete when
# the gradient of the trunk output x1 has been formed (autograd runs later-created nodes first).
LATE_PREFIXES = ('encoder.resnet.', 'encoder.e_conv1.', 'encoder.pointnet_plus.')


class Trainer:
    """train(epoch, loader) / train_step(batch) for `HandNET_GCN` + `CtdetLoss`."""

    def __init__(self, opt, model, loss, lr=1e-4, use_graph=False):
        self.opt = opt
        self.model_with_loss = ModleWithLoss(model, loss)
        named = list(model.named_parameters())
        early = [p for n, p in named if not n.startswith(LATE_PREFIXES)]
        late = [p for n, p in named if n.startswith(LATE_PREFIXES)]
        self.optimizer = FlatAdam(early + late, lr=lr)         # flat order: [early | late], see LATE_PREFIXES
        self.n_early = self.optimizer.offsets[len(early)] if late else self.optimizer.numel
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self._early_works = None
        self.force_collectives = False                         # tests: run the all-reduces on a one-rank group as well
        self.collectives = True                                # False: a rank-local step (bench.py's instrumented step on rank 0)
        self.early_probe = None                                # test hook: called with the early gradient slice when it is complete
        if hasattr(model, 'encoder'):
            model.encoder.on_trunk_output_grad = self._early_grads_ready
        self.use_graph = use_graph
        self._graph = None
        self._static = None
        self._loss_out = None

    def _early_grads_ready(self):
        """Runs inside the backward, when d loss / d x1 is complete: start the all-reduce of the early part of the flat
        gradient buffer while the trunk's backward (~a third of the step) still runs."""
        if self.use_graph or self._early_works is not None or not self.collectives or \
                (self.world == 1 and self.early_probe is None and not self.force_collectives):
            return
        F.join_wgrad()                                         # the side-stream kernels issued so far wrote into this part
        early = self.optimizer.flat_g[:self.n_early]
        if self.early_probe is not None:
            self.early_probe(early)
        self._early_works = allreduce_flat_grads(early, chunks=3, wait=False, force=self.force_collectives)

    def _fwd_bwd(self, batch, epoch):
        self._early_works = None
        self.optimizer.zero_grad()
        loss, stats, _, _ = self.model_with_loss(batch, 'train', epoch)
        loss = loss.mean()                                     # base_trainer.py:144
        loss.backward()
        F.join_wgrad()                                         # side-stream weight-gradient kernels -> flat_g complete
        F.step_counter(loss.device).add_(1)                    # fresh dropout masks next step (also under replay)
        return loss.detach(), stats

    def train_step(self, batch, epoch=0):
        """batch: dict of device tensors. Returns the (device) scalar loss; no host sync."""
        self.model_with_loss.train()
        if self.use_graph:
            loss = self._graph_step(batch, epoch)
        else:
            loss, _ = self._fwd_bwd(batch, epoch)
        if not self.collectives:
            pass
        elif self._early_works is not None:                    # early part already in flight (or nothing to do at world 1)
            allreduce_flat_grads(self.optimizer.flat_g[self.n_early:], chunks=2, force=self.force_collectives)
            for w in self._early_works:
                w.wait()
        else:
            allreduce_flat_grads(self.optimizer.flat_g, force=self.force_collectives)
        self.optimizer.step(grad_scale=1.0 / self.world if self.collectives else 1.0)
        return loss

    def _graph_step(self, batch, epoch):
        if self._graph is None:
            self._static = {k: v.clone() for k, v in batch.items()}
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):                         # warm-up on a side stream (allocator, lazy init)
                for _ in range(2):
                    self._fwd_bwd(self._static, epoch)
            torch.cuda.current_stream().wait_stream(s)
            self._graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph):
                self._loss_out, _ = self._fwd_bwd(self._static, epoch)
        for k, v in batch.items():
            self._static[k].copy_(v, non_blocking=True)
        self._graph.replay()
        return self._loss_out

    def train(self, epoch, loader, device):
        tot, n = 0.0, 0
        for batch in loader:
            batch = {k: v.to(device, non_blocking=True) for k, v in batch.items()}
            loss = self.train_step(batch, epoch)
            tot, n = tot + float(loss), n + 1                 # one host sync per iteration, like the reference's logging
        return tot / max(n, 1)


def init_distributed():
    """One process per GPU; RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* from the environment (torch.distributed.run).
    backend 'nccl' is RCCL on ROCm (the reference's string, main.py:69)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        torch.cuda.set_device(local)
        dist.init_process_group('nccl', init_method='env://', rank=rank, world_size=world)
    else:
        torch.cuda.set_device(local)
    return rank, local, world


def mpjpe_mm(pred, gt):
    """base_trainer.py:263-285: mean ||pred - gt||_2 * 1000."""
    return torch.norm(pred - gt, dim=-1).mean() * 1000
