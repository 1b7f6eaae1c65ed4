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
import logging
import os
import re

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
        self.late_join = False                                 # Trainer sets it: it joins after the backward instead
        if hasattr(model, 'join_deferred') and os.environ.get('PDFNET_DEFER_MID', '1') != '0':
            model.defer_mid_model = True                       # joined below, right after the loss has been issued

    def forward(self, batch, mode, epoch):
        ind = batch['ind'] if mode == 'train' else None
        result, paramsDict, handDictList, otherInfo = self.model(
            batch['input'], batch['choose'], batch['cloud'], batch.get('depth'), ind, batch['K_new'], batch['valid'])
        out = self.loss(result, paramsDict, handDictList, otherInfo, batch, mode, epoch)
        if hasattr(self.model, 'run_deferred_heads'):
            self.model.run_deferred_heads()                    # the loss-less wh / params heads, when the trainer deferred them (beside the backward)
        # mid_model's deferred pass (dead outputs, live BatchNorm statistics) is joined by whoever reads or writes that state next: the
        # trainer after the backward (`late_join`: round 5 -- joined here it sat between the loss and the backward, 0.5 ms of the chain),
        # everybody else right here; the model's own forward / state_dict / train() join it in any case
        if hasattr(self.model, 'join_deferred') and not (self.late_join and mode == 'train'):
            self.model.join_deferred()
        return out


class FlatAdam:
    """torch.optim.Adam(params, lr) semantics (main.py:63) on flat buffers with one fused HIP kernel.

    `params` fixes the order of the flat buffers; `ckpt_order` (default: the same list) is the order the reference's
    optimizer sees -- `model.parameters()` -- and is the index space of `state_dict()`, which is written and read in
    torch.optim.Adam's own format ({'state': {i: {'step', 'exp_avg', 'exp_avg_sq'}}, 'param_groups': [...]}) so that
    `load_model(..., resume=True)` (lib/utils/utils.py:84-96) works across the two implementations in both directions.
    `n_live` (flat offset): the range [n_live, numel) holds tensors that never receive a gradient; the Adam launch and the
    gradient all-reduce stop there (their moments stay zero, so skipping them is exact)."""

    def __init__(self, params, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, ckpt_order=None, n_live_params=None):
        self.params = [p for p in params]
        dev = self.params[0].device
        assert dev.type == 'cuda', "FlatAdam: parameters must be on the GPU (no CPU fallback)"
        self.param_groups = [{'lr': lr, 'betas': tuple(betas), 'eps': eps, 'weight_decay': 0, 'amsgrad': False}]
        ALIGN = 64                                     # every tensor starts on a 256-byte boundary: the GEMM fast
        offs, n = [], 0                                # path needs 16-byte aligned operands (float4 loads)
        for p in self.params:
            offs.append(n)
            n += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        self.numel = n
        self.offsets = offs                            # flat offset of params[i] (the buffers follow the order given)
        self.n_live = n if n_live_params is None else (offs[n_live_params] if n_live_params < len(offs) else n)
        self.flat_p = torch.zeros(n, device=dev)
        self.flat_g = torch.zeros(n, device=dev)
        self.flat_m = torch.zeros(n, device=dev)
        self.flat_v = torch.zeros(n, device=dev)
        self._grad_views = []
        for p, o in zip(self.params, offs):
            k = p.numel()
            view = self.flat_p[o:o + k].as_strided(p.shape, p.stride())      # keeps e.g. channels_last storage
            view.copy_(p.data)
            p.data = view
            p.grad = self.flat_g[o:o + k].as_strided(p.shape, p.stride())
            self._grad_views.append(p.grad)
            p._pdf_main_grad = True                   # HIP backward kernels accumulate straight into this view
        pos = {id(p): i for i, p in enumerate(self.params)}
        order = self.params if ckpt_order is None else [p for p in ckpt_order]
        assert len(order) == len(self.params) and all(id(p) in pos for p in order), "FlatAdam: ckpt_order must be a permutation of params"
        self.ckpt_index = [pos[id(p)] for p in order]                        # checkpoint index -> position in the flat order
        self.step_t = torch.zeros(1, device=dev)                             # device-side step count (graph-safe)
        self._p16_synced = False
        self.corr = torch.ones(2, device=dev)
        self._b = torch.tensor(betas, device=dev)

    # the reference edits `param_group['lr']` in place (lib/utils/utils.py:93-94, main.py:129-132)
    @property
    def lr(self):
        return self.param_groups[0]['lr']

    @lr.setter
    def lr(self, v):
        if v is not None:
            self.param_groups[0]['lr'] = float(v)

    @property
    def betas(self):
        return self.param_groups[0]['betas']

    @property
    def eps(self):
        return self.param_groups[0]['eps']

    def refresh_bf16_shadows(self):
        """bf16 mode: one cast of the flat master buffer into a flat bf16 buffer whose per-parameter views (same shapes and
        strides) ride on the parameters as their shadows (pdfnet_amd.functional.shadow_of): the GEMM kernels then read 2-byte
        weights.  Called at the start of every train step, i.e. after whatever changed the parameters last."""
        if getattr(self, 'flat_p16', None) is None:
            self.flat_p16 = torch.empty(self.numel, dtype=torch.bfloat16, device=self.flat_p.device)
            self._p16_views = [self.flat_p16[o:o + p.numel()].as_strided(p.shape, p.stride()) for p, o in zip(self.params, self.offsets)]
        hip.lib().pdf_cast_bf16(hip.ptr(self.flat_p), hip.ptr(self.flat_p16), self.numel, hip.stream())
        self._cast_transposed()
        base = self.flat_p.data_ptr()
        for p, v, o in zip(self.params, self._p16_views, self.offsets):
            if p.data_ptr() == base + 4 * o:                  # still the view into the flat buffer
                F.attach_shadow(p, v)
        if getattr(self, 'flat_p16t', None) is not None:
            for i, v in self._t_views:
                p = self.params[i]
                p._pdf_bf16_t = v if p.data_ptr() == base + 4 * self.offsets[i] else None
        self._p16_synced = True

    def set_transposed(self, entries):
        """entries: [(parameter, R, T, C)] -- tensors that also get a TRANSPOSED bf16 shadow (bf16 mode; see refresh_bf16_shadows)."""
        pos = {id(p): i for i, p in enumerate(self.params)}
        self._t_entries = [(pos[id(p)], int(R), int(T), int(C)) for p, R, T, C in entries if id(p) in pos]
        self.flat_p16t = None

    def _cast_transposed(self):
        """wt[C][T][R] (bf16) of every listed weight w[R][T][C], one launch (pdf_cast_bf16_transposed) -- the operand the LDS-DMA
        kernel reads in a backward-data GEMM (PdfCallOpts::op1_bf16_t)."""
        ent = getattr(self, '_t_entries', None)
        if not ent or not F.TRANSPOSED_SHADOWS:
            return
        if self.flat_p16t is None:
            import numpy as np
            self.flat_p16t = torch.empty(self.numel, dtype=torch.bfloat16, device=self.flat_p.device)
            tab = np.zeros(len(ent), dtype=np.dtype([('src', '<i8'), ('dst', '<i8'), ('R', '<i4'), ('T', '<i4'), ('C', '<i4'), ('tile0', '<i4')]))
            tiles = 0
            for k, (i, R, T, C) in enumerate(ent):
                tab[k] = (self.offsets[i], self.offsets[i], R, T, C, tiles)
                tiles += T * ((R + 31) // 32) * ((C + 31) // 32)
            self._t_tiles = tiles
            self._t_table = torch.from_numpy(tab.view(np.uint8).copy()).to(self.flat_p.device)
            self._t_views = [(i, self.flat_p16t[self.offsets[i]:self.offsets[i] + self.params[i].numel()]) for i, _, _, _ in ent]
        hip.lib().pdf_cast_bf16_transposed(hip.ptr(self.flat_p), hip.ptr(self.flat_p16t), hip.ptr(self._t_table), len(ent), self._t_tiles, hip.stream())

    def params_changed(self):
        """Something other than `step()` wrote the parameters (checkpoint load, replica broadcast, user code): the next train
        step re-casts and re-attaches the bf16 shadows."""
        self._p16_synced = False

    def reattach_grads(self):
        """The HIP backward kernels accumulate into `p.grad` only while it is the view into the flat gradient buffer.
        `model.zero_grad()` (set_to_none=True), `p.grad = None` or an optimizer-style zero_grad replace it; autograd would
        then allocate fresh gradients that step() never reads.  Called before every backward: puts the views back."""
        for p, v in zip(self.params, self._grad_views):
            if p.grad is not v:
                p.grad = v

    def zero_grad(self):
        self.reattach_grads()
        self.flat_g[:self.n_live].zero_()

    def begin_step(self):
        """Advance the step count and the bias corrections (device side) -- once per optimizer step, before any `apply`."""
        self.step_t += 1
        torch.sub(1.0, torch.pow(self._b, self.step_t), out=self.corr)        # [1-b1^t, 1-b2^t]

    def apply(self, lo, hi, grad_scale=1.0, stream=None):
        """The Adam update of the flat range [lo, hi) on `stream` (default: the current one)."""
        if hi <= lo:
            return
        off = 4 * lo
        hip.lib().pdf_adam_step(self.flat_p.data_ptr() + off, self.flat_g.data_ptr() + off, self.flat_m.data_ptr() + off,
                                self.flat_v.data_ptr() + off, hi - lo, float(self.lr), self.betas[0], self.betas[1], self.eps,
                                hip.ptr(self.corr), float(grad_scale), hip.stream() if stream is None else stream)

    def step(self, grad_scale=1.0, done_upto=0):
        """done_upto > 0: the range [0, done_upto) has been updated already this step (Trainer: the early slice, on a side stream
        beside the trunk backward, after `begin_step`)."""
        F.join_wgrad()                                 # side-stream gradient kernels must have landed (no-op when already joined)
        if done_upto <= 0:
            self.begin_step()
        self.apply(done_upto, self.n_live, grad_scale)
        if getattr(self, 'flat_p16', None) is not None and F.shadows_on():
            # The update went through raw pointers (no version bump on the parameters), so the attached shadows -- views into
            # flat_p16 -- would go on serving the PREVIOUS weights to whatever runs next (evaluation, a plain model call):
            # re-cast in place right here; the views stay attached and are current again.
            hip.lib().pdf_cast_bf16(hip.ptr(self.flat_p), hip.ptr(self.flat_p16), self.n_live, hip.stream())
            if getattr(self, 'flat_p16t', None) is not None:
                self._cast_transposed()
        else:
            # fp32 mode (or shadows off): nothing re-cast, so a bf16 buffer from an earlier bf16 phase is stale from here on; the
            # next bf16 train step / evaluation refreshes it (toggling the GEMM precision on a live Trainer).  The
            # attached views are taken off the parameters -- once, on the first such step -- so that a plain model call made
            # after switching back to bf16 converts the fp32 master weights instead of multiplying with the old copy.
            if self._p16_synced:
                for p in self.params:
                    if getattr(p, '_pdf_bf16', None) is not None:
                        p._pdf_bf16 = None
                    if getattr(p, '_pdf_bf16_t', None) is not None:
                        p._pdf_bf16_t = None
            self._p16_synced = False

    def _view(self, flat, i):
        p, o = self.params[i], self.offsets[i]
        return flat[o:o + p.numel()].as_strided(p.shape, p.stride())

    def state_dict(self):
        """torch.optim.Adam's layout: per-parameter moments in logical (OIHW-contiguous) form, indexed in `ckpt_order`."""
        step = self.step_t.detach().cpu().reshape(())
        state = {}
        for ci, i in enumerate(self.ckpt_index):
            state[ci] = {'step': step.clone(), 'exp_avg': self._view(self.flat_m, i).detach().cpu().contiguous(),
                         'exp_avg_sq': self._view(self.flat_v, i).detach().cpu().contiguous()}
        g = dict(self.param_groups[0])
        g.update({'maximize': False, 'foreach': None, 'capturable': False, 'differentiable': False, 'fused': None,
                  'params': list(range(len(self.ckpt_index)))})
        return {'state': state, 'param_groups': [g]}

    def load_state_dict(self, sd):
        if 'state' not in sd or 'param_groups' not in sd:
            raise ValueError("FlatAdam.load_state_dict: expected torch.optim.Adam's {'state', 'param_groups'} layout")
        ids = sd['param_groups'][0]['params']
        if len(ids) != len(self.ckpt_index):
            raise ValueError("FlatAdam.load_state_dict: checkpoint has %d parameters, the model %d" % (len(ids), len(self.ckpt_index)))
        self.flat_m.zero_()
        self.flat_v.zero_()
        step = 0.0
        for ci, pid in enumerate(ids):
            st = sd['state'].get(pid)
            if st is None:                             # torch keeps no state for a parameter that never had a gradient
                continue
            i = self.ckpt_index[ci]
            if tuple(st['exp_avg'].shape) != tuple(self.params[i].shape):
                raise ValueError("FlatAdam.load_state_dict: parameter %d has shape %s in the checkpoint, %s in the model"
                                 % (ci, tuple(st['exp_avg'].shape), tuple(self.params[i].shape)))
            self._view(self.flat_m, i).copy_(st['exp_avg'])
            self._view(self.flat_v, i).copy_(st['exp_avg_sq'])
            step = max(step, float(st['step']))
        self.step_t.fill_(step)
        g = sd['param_groups'][0]
        self.param_groups[0].update({'lr': float(g['lr']), 'betas': tuple(g['betas']), 'eps': g['eps']})
        self._b.copy_(torch.tensor(self.betas))


class _Reduction:
    """One outstanding gradient all-reduce; `finish()` waits for it (and, when the gradients travel in a narrower dtype,
    widens the reduced values back into the flat buffer)."""

    def __init__(self, work, dst=None, buf=None):
        self.work, self.dst, self.buf = work, dst, buf

    def finish(self):
        self.work.wait()
        if self.buf is not None:
            self.dst.copy_(self.buf)


def allreduce_flat_grads(flat_g, chunks=4, wait=True, force=False, comm_dtype=None):
    """Gradient SUM across ranks (the mean's 1/world is folded into the optimizer's grad_scale).  A few large
    RCCL all-reduces on the contiguous buffer: xGMI rings are per-link bound, so fewer/larger beats many/small.
    wait=False returns the outstanding reductions instead of finishing them; force=True issues the collectives on
    a one-rank group too (tests: exercises the RCCL calls on a one-GPU box).  comm_dtype=torch.bfloat16 sends the gradients
    as bf16 (half the bytes over xGMI; BASELINE configs 4-5) and accumulates the result back into the fp32 buffer."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force) or flat_g.numel() == 0:
        return []
    n = flat_g.numel()
    per = (n + chunks - 1) // chunks
    reds = []
    for i in range(0, n, per):
        piece = flat_g[i:min(n, i + per)]
        if comm_dtype is not None and comm_dtype != piece.dtype:
            buf = piece.to(comm_dtype)
            reds.append(_Reduction(dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True), piece, buf))
        else:
            reds.append(_Reduction(dist.all_reduce(piece, op=dist.ReduceOp.SUM, async_op=True)))
    if not wait:
        return reds
    for r in reds:
        r.finish()
    return []


class GradReducer:
    """The data-parallel gradient reduction of one step over the flat gradient buffer laid out [early | late | never used]:
    `early_ready()` is called from inside the backward as soon as the early slice is complete (autograd hook on the trunk
    output) and starts its all-reduce while the rest of the backward runs; `finish()` is called after the backward, reduces
    the late slice and waits for everything.  The never-used tail [n_live, numel) is not sent.  Device-agnostic (the CPU
    tests drive it over gloo; the trainer over RCCL)."""

    def __init__(self, flat_g, n_early, n_live, comm_dtype=None):
        self.flat_g, self.n_early, self.n_live, self.comm_dtype = flat_g, n_early, n_live, comm_dtype
        self.force = False
        self.early = None
        self.bytes_sent = 0

    def reset(self):
        self.early = None

    def early_ready(self):
        if self.early is not None:
            return
        self.early = allreduce_flat_grads(self.flat_g[:self.n_early], chunks=3, wait=False, force=self.force, comm_dtype=self.comm_dtype)

    def finish(self):
        live = self.flat_g[:self.n_live]
        esz = torch.empty(0, dtype=self.comm_dtype or live.dtype).element_size()
        self.bytes_sent = live.numel() * esz
        if self.early is not None:                            # early part already in flight
            allreduce_flat_grads(live[self.n_early:], chunks=2, force=self.force, comm_dtype=self.comm_dtype)
            for r in self.early:
                r.finish()
        else:
            allreduce_flat_grads(live, force=self.force, comm_dtype=self.comm_dtype)


# Parameters whose gradients are complete only at the very end of the backward: the ResNet trunk and what hangs off the
# stem (e_conv1, and PointNet++ levels 1-2, whose backward runs last on its side stream).  Everything else -- two thirds
# of the buffer: pyramid laterals, feat, heads, dense decoders, centre convolutions, SFT, mesh decoder -- is complete when
# the gradient of the trunk output x1 has been formed (autograd runs later-created nodes first).
LATE_PREFIXES = ('encoder.resnet.', 'encoder.e_conv1.', 'encoder.pointnet_plus.')

# Parameters no autograd path of HandNET_GCN.forward reaches, whatever the loss (the reference needs
# find_unused_parameters=True for them, base_trainer.py:95): constructed-but-unused heads, the mid model (run under
# no_grad for its BN statistics), the image-feature extractors of the dual GCN and the GCN blocks' overwritten norm1
# (SURVEY Appendix A.1).  13.6 M elements: laid out BEHIND the live ranges so that zero_grad, the gradient all-reduce and
# Adam stop before them.  tests/test_host_cpu.py checks the rule against tests/golden/params_without_grad.txt.
DEAD_PATTERN = re.compile(r'^(mid_model\.|encoder\.(joint_head_l|joint_head_r|mano_head|resnet\.fc|pointnet_plus\.netR_FC)\.|'
                          r'decoder\.dual_gcn\.layers\.\d+\.(img_ex_(left|right)\.|graph_(left|right)\.GCN_blocks\.\d+\.norm1\.))')


def split_parameters(named):
    """-> (early, late, dead) lists of (name, parameter), each in model order."""
    early, late, dead = [], [], []
    for n, p in named:
        (dead if DEAD_PATTERN.match(n) else late if n.startswith(LATE_PREFIXES) else early).append((n, p))
    return early, late, dead


def _transposed_entries(model):
    """[(weight parameter, R, T, C)] for the conv / linear layers whose backward-data GEMM is worth handing to the LDS-DMA bf16 kernel
    (pdf_cast_bf16_transposed: w[R][T][C] -> wt[C][T][R]; conv weights are stored channels_last = [Cout][KH][KW][Cin])."""
    from ..networks import layers
    out = []
    for m in model.modules():
        w = getattr(m, 'weight', None)
        if isinstance(m, layers.Conv2d) and w.is_contiguous(memory_format=torch.channels_last):
            R, C, T = w.shape[0], w.shape[1], w.shape[2] * w.shape[3]
        elif isinstance(m, layers.Linear) and w.is_contiguous():
            R, C, T = w.shape[0], w.shape[1], 1
        else:
            continue
        if R % 8 == 0 and C % 8 == 0 and w.numel() >= 16384:
            out.append((w, R, T, C))
    return out


class Trainer:
    """train(epoch, loader) / train_step(batch) for `HandNET_GCN` + `CtdetLoss`."""

    def __init__(self, opt, model, loss, lr=1e-4, use_graph=False, broadcast_buffers=False, grad_comm_dtype=None):
        self.opt = opt
        self.model = model
        self.model_with_loss = ModleWithLoss(model, loss)
        self.model_with_loss.late_join = os.environ.get('PDFNET_MID_LATE_JOIN', '1') != '0'
        # the wh / params heads have no term in CtdetLoss (lib/trains/simplified.py:397-399): under THIS loss they are issued after it and joined
        # with mid_model after the backward; any other loss module sees them computed in place
        from .simplified import CtdetLoss as _Ctdet
        if hasattr(model, 'defer_lossless_heads') and isinstance(loss, _Ctdet) and self.model_with_loss.late_join:
            # measured neutral (627 vs 628 img/s: the forward's mesh-level window shrinks 3.4 -> 2.5 ms, the backward's first windows grow by as
            # much -- the step conserves work, the heads' 0.9 ms of convolutions just move): opt-in
            model.defer_lossless_heads = os.environ.get('PDFNET_DEFER_HEADS', '0') != '0'
        named = list(model.named_parameters())
        early, late, dead = split_parameters(named)
        flat = [p for _, p in early + late + dead]             # flat order: [early | late | never used]
        self.optimizer = FlatAdam(flat, lr=lr, ckpt_order=[p for _, p in named], n_live_params=len(early) + len(late))
        self.optimizer.set_transposed(_transposed_entries(model))   # bf16 mode: transposed shadows of the large conv / linear weights
        self.n_early = self.optimizer.offsets[len(early)] if late else self.optimizer.n_live
        self.n_live = self.optimizer.n_live
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank() if self.world > 1 else 0
        self.reducer = GradReducer(self.optimizer.flat_g, self.n_early, self.n_live, grad_comm_dtype)
        self.force_collectives = False                         # tests: run the all-reduces on a one-rank group as well
        self.collectives = True                                # False: a rank-local step (bench.py's instrumented step on rank 0)
        self.early_probe = None                                # test hook: called with the early gradient slice when it is complete
        # DDP(broadcast_buffers=True) re-sends rank 0's BatchNorm statistics before every forward (base_trainer.py:94-95).
        # Off by default: statistics then evolve per rank and rank 0's are what a checkpoint holds.
        self.broadcast_buffers = broadcast_buffers
        if hasattr(model, 'encoder'):
            model.encoder.on_trunk_output_grad = self._early_grads_ready
        # use_graph: False = eager launches (forked side streams: the default, fastest where the GPU is the limit), True = the whole
        # forward + loss + backward replayed as one hipGraph (single stream: loses the weight-gradient overlap but costs no host
        # time), 'auto' = time a few steps of each at the start and keep the faster one.  Measured on one MI355X: fp32 B=32
        # eager 470 vs graph 442 img/s; fp32 B=8 199 vs 271; bf16 B=32 803 vs 840 on a slow host -- the host-bound regimes win.
        self._auto = None
        if use_graph == 'auto':
            if self.world > 1:
                use_graph = False                              # (the overlapped gradient all-reduce lives in the eager backward)
            else:
                use_graph, self._auto = False, {'phase': 'eager', 'marks': [], 'eager_ms': None}
        self.use_graph = use_graph
        self._graphs = {}
        self._modes_logged = False
        self._graph_pool = None
        # (measured on MI355X: 446 vs 450 img/s -- the 2.4 GB Adam pass takes HBM bandwidth and CUs from the trunk backward's
        # BatchNorm / GEMM kernels; off by default)
        self.overlap_adam = os.environ.get('PDFNET_OVERLAP_ADAM', '0') != '0'
        self._adam_stream, self._adam_done, self._in_train_step = None, 0, False
        self._comm_stream = None                               # the early gradient all-reduce is issued from here (_early_grads_ready)
        hip.check_device(self.optimizer.flat_p.device)
        model.register_load_state_dict_post_hook(lambda *_: self.optimizer.params_changed())
        if self.world > 1:
            self.sync_replicas()

    def sync_replicas(self):
        """What DistributedDataParallel's constructor does: every rank starts from rank 0's parameters and buffers."""
        dist.broadcast(self.optimizer.flat_p, 0)
        self.optimizer.params_changed()
        for b in self._float_buffers():
            dist.broadcast(b, 0)

    def _float_buffers(self):
        return [b for b in self.model.buffers() if b.is_floating_point() and b.numel() > 0]

    def _early_grads_ready(self):
        """Runs inside the backward, when d loss / d x1 is complete: start the all-reduce of the early part of the flat
        gradient buffer while the trunk's backward (~a third of the step) still runs."""
        if self.world == 1 and self.early_probe is None and not self.force_collectives and not self.use_graph and self.collectives \
                and self.overlap_adam and self._adam_done == 0 and self._in_train_step:
            # One GPU, nothing to reduce: the early two thirds of the parameters have their final gradients now, so their Adam
            # update (HBM-bound, 0.3 ms) runs on a side stream beside the trunk's backward (MFMA-bound) instead of after it.
            # The side stream waits for everything issued so far on this stream and on the weight-gradient streams.
            dev = hip._raw_device()
            cur = hip._raw_stream(dev)
            if self._adam_stream is None:
                self._adam_stream = torch.cuda.Stream()
            side = self._adam_stream.cuda_stream
            self.optimizer.begin_step()
            F.flush_wgrad()
            wait = hip.lib().pdf_stream_wait
            wait(side, cur)
            for key in list(F._wg_used):
                wait(side, F._wg_streams[key][1])
            self.optimizer.apply(0, self.n_early, 1.0, stream=side)
            self._adam_done = self.n_early
            return
        if self.use_graph or self.reducer.early is not None or not self.collectives or \
                (self.world == 1 and self.early_probe is None and not self.force_collectives):
            return
        if self.early_probe is not None:                       # (test hook: it reads the slice on THIS stream, so this stream joins)
            F.join_wgrad(keep=True)
            self.early_probe(self.optimizer.flat_g[:self.n_early])
        self.reducer.force = self.force_collectives
        if not self.optimizer.flat_g.is_cuda:
            self.reducer.early_ready()
            return
        # The early slice was written by the weight-gradient side streams (and by a few bias gradients on this stream).  Round 4
        # joined those streams into THIS stream here, which parked the trunk's backward -- the third of the step the all-reduce is
        # supposed to hide under -- behind everything they still held.  Now a dedicated communication stream
        # waits for them; ProcessGroupNCCL orders its own stream after the stream that is current when the collective is issued,
        # so the all-reduce (and the bf16 cast in front of it) starts when the gradients are complete and this stream never waits.
        dev = hip._raw_device()
        cur = hip._raw_stream(dev)
        if self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream()
        comm = self._comm_stream.cuda_stream
        F.flush_wgrad()                                        # (grouped weight gradients: the early slice's last group goes out now)
        wait = hip.lib().pdf_stream_wait
        wait(comm, cur)
        for key in list(F._wg_used):
            wait(comm, F._wg_streams[key][1])
        with torch.cuda.stream(self._comm_stream):
            self.reducer.early_ready()
        for r in self.reducer.early or ():                     # staged on the communication stream, finished (copied back) on this one
            if r.buf is not None:
                r.buf.record_stream(torch.cuda.current_stream())

    def _fwd_bwd(self, batch, epoch):
        self.reducer.reset()
        self.optimizer.zero_grad()
        loss, stats, _, _ = self.model_with_loss(batch, 'train', epoch)
        loss = loss.mean()                                     # base_trainer.py:144
        loss.backward()
        if hasattr(self.model, 'join_deferred'):
            self.model.join_deferred()                         # mid_model's BatchNorm-statistics pass (see ModleWithLoss.forward)
        F.join_wgrad()                                         # side-stream weight-gradient kernels -> flat_g complete
        F.step_counter(loss.device).add_(1)                    # fresh dropout masks next step (also under replay)
        return loss.detach(), stats

    def train_step(self, batch, epoch=0):
        """batch: dict of device tensors. Returns the (device) scalar loss; no host sync."""
        self.model_with_loss.train()
        if not self._modes_logged:                             # once per run: what the 'auto' switches resolved to for this batch size
            self._modes_logged = True
            b = next((v.shape[0] for v in batch.values() if torch.is_tensor(v) and v.dim() > 0), 0)
            self.resolved_modes = dict(F.bf16_modes(b), batch=b, world=self.world, launch='hipGraph' if self.use_graph is True else ('auto' if self._auto is not None else 'eager'),
                                       mesh_decoder='fused' if F.MESH_FUSED else 'per-op', mesh_loss='fused' if F.MESH_LOSS_FUSED else 'per-term')
            if self.rank == 0:
                logging.getLogger('pdfnet_amd').info("trainer modes: %s", self.resolved_modes)
        if F.shadows_on() and not self.optimizer._p16_synced:
            self.optimizer.refresh_bf16_shadows()              # (after a step() the shadows are re-cast in place: nothing to do)
        if self.broadcast_buffers and self.world > 1 and self.collectives:
            for b in self._float_buffers():
                dist.broadcast(b, 0)
        self._adam_done, self._in_train_step = 0, True
        try:
            if self.use_graph:
                loss = self._graph_step(batch, epoch)
            else:
                loss, _ = self._fwd_bwd(batch, epoch)
        finally:
            self._in_train_step = False
        if self.collectives:                                   # the never-used tail is not reduced (zero on every rank)
            self.reducer.force = self.force_collectives
            self.reducer.finish()
        if self._adam_done:                                    # the early slice was updated beside the trunk backward
            hip.lib().pdf_stream_wait(hip.stream(), self._adam_stream.cuda_stream)
        self.optimizer.step(grad_scale=1.0 / self.world if self.collectives else 1.0, done_upto=self._adam_done)
        self._adam_done = 0
        if self._auto is not None:
            self._auto_select()
        return loss

    GRAPH_WGRAD_GROUP = int(os.environ.get('PDFNET_GRAPH_WGRAD_GROUP', '16'))
    AUTO_STEPS = 4                                             # timed steps per mode ...
    AUTO_SKIP = 3                                              # ... after this many untimed ones (allocator warm-up; capture)

    def _auto_select(self):
        """use_graph='auto': an event after every optimizer step; AUTO_SKIP + AUTO_STEPS + 1 eager steps, then as many graph steps (the
        first AUTO_SKIP of each mode are not counted: allocator warm-up / capture), one synchronisation, and the mode with the lower median step
        time stays.  A stream of batches whose shapes change keeps re-capturing; such loaders should pass use_graph=False."""
        a = self._auto
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        a['marks'].append(e)
        if len(a['marks']) < self.AUTO_SKIP + self.AUTO_STEPS + 1:
            return
        torch.cuda.synchronize()
        m = a['marks']
        ms = sorted(m[i].elapsed_time(m[i + 1]) for i in range(self.AUTO_SKIP, len(m) - 1))
        med = ms[len(ms) // 2]
        if a['phase'] == 'eager':
            a['eager_ms'], a['phase'], a['marks'] = med, 'graph', []
            self.use_graph = True
            return
        self.auto_choice = {'eager_ms': a['eager_ms'], 'graph_ms': med}
        if med >= a['eager_ms']:                               # eager stays: give the captured graphs and their pool back
            self.use_graph = False
            self._graphs.clear()
            self._graph_pool = None
        self._auto = None

    def _graph_step(self, batch, epoch):
        """One hipGraph per (loss schedule phase, batch signature): `epoch` only enters the step through
        alpha = 0 if epoch < 20 else 1 (simplified.py:610), which switches two loss weights and the edge-length gradient,
        so a graph captured before epoch 20 must not be replayed after it; a last, smaller batch gets its own graph too."""
        tens = {k: v for k, v in batch.items() if torch.is_tensor(v)}
        key = (epoch >= 20,) + tuple((k, tuple(v.shape), v.dtype) for k, v in sorted(tens.items()))
        entry = self._graphs.get(key)
        if entry is None:
            static = {k: v.clone() for k, v in tens.items()}
            # The two un-captured warm-up passes (allocator, lazy init) run the model in train mode: each would apply one
            # more BatchNorm running-statistics update, bump num_batches_tracked and advance the dropout step counter without
            # an optimizer step -- state the reference advances exactly once per batch.  Snapshot and restore it.
            from ..networks.layers import BatchNorm
            BatchNorm.flush_counters()
            bufs = [b for b in self.model.buffers()]
            saved = [b.clone() for b in bufs]
            ctr = F.step_counter(self.optimizer.flat_p.device)
            ctr_saved = ctr.clone()
            # inside a captured graph every cross-stream edge costs ~14 us of the main chain (profiles/r05_hipgraph_branches.txt): the
            # weight gradients leave for their side stream GRAPH_WGRAD_GROUP at a time instead of one by one
            group_saved, F.WGRAD_GROUP = F.WGRAD_GROUP, max(F.WGRAD_GROUP, self.GRAPH_WGRAD_GROUP)
            try:
                s = torch.cuda.Stream()
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):                     # warm-up on a side stream
                    for _ in range(2):
                        self._fwd_bwd(static, epoch)
                    BatchNorm.flush_counters()
                torch.cuda.current_stream().wait_stream(s)
                for b, v in zip(bufs, saved):
                    b.copy_(v)
                ctr.copy_(ctr_saved)
                graph = torch.cuda.CUDAGraph()
                if self._graph_pool is None:
                    self._graph_pool = torch.cuda.graph_pool_handle()     # one private pool for every key (not one multi-GB pool per key)
                with torch.cuda.graph(graph, pool=self._graph_pool):
                    loss_out, _ = self._fwd_bwd(static, epoch)
            finally:
                F.WGRAD_GROUP = group_saved
            entry = self._graphs[key] = (graph, static, loss_out)
        graph, static, loss_out = entry
        for k, v in tens.items():
            static[k].copy_(v, non_blocking=True)
        graph.replay()
        return loss_out

    def train(self, epoch, loader, device):
        tot, n = 0.0, 0
        if hasattr(loader, 'set_epoch'):
            loader.set_epoch(epoch)                            # main.py:108 (`ShardedLoader`: this rank's shard of this epoch)
        for batch in loader:
            # the reference loader also yields non-tensor entries ('meta', base_trainer.py:134-136 skips them)
            batch = {k: (v.to(device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in batch.items()}
            loss = self.train_step(batch, epoch)
            tot, n = tot + float(loss), n + 1                 # one host sync per iteration, like the reference's logging
        return tot / max(n, 1)


    def evaluation(self, loader, device=None, score_path=None, json_path=None):
        """Counterpart of `BaseTrainer.evaluation` (lib/trains/base_trainer.py:207-429, H2O branch): the test-mode pass
        (centres from the predicted heat-map, root from the predicted depth) and the mean Euclidean errors per hand --
        absolute and root-relative joints / vertices in mm, 2-D landmarks in pixels.  The reference evaluates on rank 0
        with batch size 1 and pulls every error to the host; here any batch size, every rank takes the batches its
        loader yields, the sums stay on the device (`F.point_dist_sum`) and are all-reduced once at the end.
        Returns a dict of floats (one host sync).

        score_path: rank 0 appends the reference's `H2O-val.txt` block (base_trainer.py:420-429).
        json_path:  rank 0 writes the reference's `hand_poses.json` submission file (base_trainer.py:328-335,432-433,486-489):
                    {"modality": "RGBD", "<action id>": {"<frame:06d>.txt": [2 x 21 x 3 absolute joints, left hand first]}};
                    needs the dataset's `id` / `frame_num` entries in every batch; the predictions of all ranks are gathered."""
        mwl = self.model_with_loss
        was_training = mwl.training
        mwl.eval()
        if F.shadows_on() and not self.optimizer._p16_synced:
            self.optimizer.refresh_bf16_shadows()
        dev = device or self.optimizer.flat_p.device
        try:
            acc = torch.zeros(11, dtype=torch.float64, device=dev)        # 5 metrics x 2 hands + sample count
            poses = []
            with torch.no_grad():
                for batch in loader:
                    batch = {k: (v.to(dev, non_blocking=True) if torch.is_tensor(v) else v) for k, v in batch.items()}
                    tup = mwl(batch, 'test', None)
                    acc += evaluation_sums(tup, batch)
                    if json_path is not None:
                        if 'id' not in batch or 'frame_num' not in batch:
                            raise KeyError("Trainer.evaluation: hand_poses.json needs batch['id'] and batch['frame_num'] (interhand.py H2O entries)")
                        jp = tup[1]
                        key = torch.stack((torch.as_tensor(batch['id'], device=dev).reshape(-1).double(),
                                           torch.as_tensor(batch['frame_num'], device=dev).reshape(-1).double()), 1)
                        poses.append(torch.cat((key, jp.reshape(jp.shape[0], -1).double()), 1))          # [B, 2 + 126]
            if self.world > 1:
                dist.all_reduce(acc)
            out = finish_evaluation(acc.cpu())
            if json_path is not None:
                rows = torch.cat(poses) if poses else torch.zeros((0, 128), dtype=torch.float64, device=dev)
                if self.world > 1:                                 # ragged gather: pad every rank's block to the longest
                    n = torch.tensor([rows.shape[0]], device=dev)
                    ns = [torch.zeros_like(n) for _ in range(self.world)]
                    dist.all_gather(ns, n)
                    m = max(int(x) for x in ns)
                    pad = torch.zeros((m, rows.shape[1]), dtype=rows.dtype, device=dev)
                    pad[:rows.shape[0]] = rows
                    parts = [torch.empty_like(pad) for _ in range(self.world)]
                    dist.all_gather(parts, pad)
                    rows = torch.cat([p[:int(k)] for p, k in zip(parts, ns)])
                if self.rank == 0:
                    write_hand_poses_json(json_path, rows.cpu())
            if score_path is not None and self.rank == 0 and out['samples'] > 0:      # (an empty loader writes no block)
                write_h2o_scores(score_path, out)
        finally:
            mwl.train(was_training)                            # restored even when a batch raises
        return out


EVAL_KEYS = ('abs_joints', 'abs_verts', 'off_joints', 'off_verts', 'lms_px')


def evaluation_sums(tup, batch):
    """test-mode 9-tuple of the loss module (simplified.py:652-653) -> float64 [11]: per metric and hand the sum over the
    batch of the per-sample mean error, then the sample count."""
    vp, jp, vg, jg, lms, vpo, jpo, vgo, jgo = tup
    lms_gt = torch.stack((batch['lms_left_gt'], batch['lms_right_gt']), 1)
    parts = []
    for pred, gt in ((jp, jg), (vp, vg), (jpo, jgo), (vpo, vgo), (lms, lms_gt)):
        parts.append((F.point_dist_sum(pred, gt) / pred.shape[-2]).sum(0))        # [B,2] -> [2]
    n = torch.full((1,), float(vp.shape[0]), device=vp.device)
    return torch.cat(parts + [n]).double()


def finish_evaluation(acc):
    """-> {'abs_left_joints' ... 'off_right_verts' in mm, 'lms_px', 'mpjpe_mm' / 'mpvpe_mm' (both hands, absolute),
    'mpjpe_off_mm' / 'mpvpe_off_mm' (root-relative), 'samples'} -- the figures base_trainer.py:420-429 prints."""
    n = float(acc[10])
    out = {'samples': int(n)}
    if n == 0:
        return out
    for i, k in enumerate(EVAL_KEYS):
        l, r = float(acc[2 * i]) / n, float(acc[2 * i + 1]) / n
        if k == 'lms_px':
            out[k] = (l + r) / 2
        else:
            kind, what = k.split('_')
            out['%s_left_%s' % (kind, what)], out['%s_right_%s' % (kind, what)] = l * 1000, r * 1000
    out['mpjpe_mm'] = (out['abs_left_joints'] + out['abs_right_joints']) / 2
    out['mpvpe_mm'] = (out['abs_left_verts'] + out['abs_right_verts']) / 2
    out['mpjpe_off_mm'] = (out['off_left_joints'] + out['off_right_joints']) / 2
    out['mpvpe_off_mm'] = (out['off_left_verts'] + out['off_right_verts']) / 2
    return out


def write_h2o_scores(path, ev):
    """Append the block `BaseTrainer.evaluation` writes to `H2O-val.txt` (base_trainer.py:420-429; same keys, order, %.2f mm)."""
    with open(path, 'a') as fo:
        fo.write('eval \n')
        for kind in ('abs', 'off'):
            for what in ('joints', 'verts'):
                for hand in ('left', 'right'):
                    fo.write('%s_%s_%s_loss_all: %.2f\n' % (kind, hand, what, ev['%s_%s_%s' % (kind, hand, what)]))


def write_hand_poses_json(path, rows):
    """rows [n, 2 + 126] = (action id, frame number, absolute joints of both hands) -> the H2O submission file the reference
    dumps at base_trainer.py:486-489: actions in ascending id, frames in the order they were evaluated."""
    import json
    out = {"modality": "RGBD"}
    for r in rows.tolist():
        out.setdefault('%d' % int(r[0]), {})['%06d.txt' % int(r[1])] = [float(x) for x in r[2:]]
    ordered = {"modality": "RGBD"}
    for k in sorted((k for k in out if k != "modality"), key=int):
        ordered[k] = out[k]
    with open(path, 'w') as fo:
        json.dump(ordered, fo)


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
    """base_trainer.py:263-285: mean ||pred - gt||_2 * 1000 over [..., n, 3] points (HIP reduction, one host sync)."""
    return float((F.point_dist_sum(pred, gt) / pred.shape[-2]).mean()) * 1000
