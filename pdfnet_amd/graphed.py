"""A chain of sub-networks replayed as hipGraphs (one forward graph; per stage a backward graph and a weight-gradient graph) inside an
otherwise eager train step.

Why (round 5): the step issues ~1,800 launches from Python in ~31 ms of host time; at fp32 B=32 the GPU needs 51 ms and the host
is never the limit, at bf16 B=32 the GPU needs ~30 ms and the host IS the limit (tools/host_time.py, profiles/r05_host_time.txt).
The whole step as ONE hipGraph costs no host time but replays slower than eager streams: a cross-stream edge inside a captured graph
costs ~14 us of the main chain on this runtime and the two sides of it barely overlap (profiles/r05_hipgraph_branches.txt), and the
step has hundreds of them.  LINEAR graphs, however, replay as fast as eager launches and are submitted in one go (~10 us of host
time for 400 kernels).  The ResNet trunk is the part of the step that is a plain chain -- 53 convolutions + 53 BatchNorms, a third
of all launches, fixed shapes, no data-dependent control flow -- so it is replayed from linear graphs only:
  forward            ONE graph, layer1..layer4;
  backward, stage k  graph A_k: the data-gradient / BatchNorm chain of layer k (main stream);
                     graph B_k: the weight gradients of layer k (the eager weight-gradient side stream, after A_k, beside A_{k-1}).
The cross-stream edges (one per stage) are eager events between graph launches, not edges inside a graph.

The mechanism is the one of torch.cuda.make_graphed_callables, restated for this library's conventions:
  * weight gradients are accumulated by the backward kernels straight into the trainer's flat gradient buffer (functional._main_grad), not
    returned to autograd -- so only Trainer-owned parameters qualify; while A_k is captured the weight-gradient launches are deferred
    (functional.WGRAD_GROUP) and then captured into B_k; their operands (activations, output gradients) stay pinned in the graphs' pool;
  * the capture's warm-up passes add garbage to those gradients and advance the BatchNorm running statistics: the former are re-zeroed
    (capture therefore happens in the forward of a step, before any real gradient exists), the latter rolled back;
  * `num_batches_tracked` is counted on the host (layers.BatchNorm._pending): a replay bumps the counters of the segment's BatchNorms itself;
  * bf16 shadows travel as Python attributes of tensors (functional.attach_shadow): the replayed outputs get theirs re-attached.
"""
import os

import torch
from torch.autograd import Function

from . import functional as F

TRUNK_GRAPH = os.environ.get("PDFNET_TRUNK_GRAPH", "0")         # '0' eager (default), '1' graphed under a Trainer
WARMUP = 2
SUSPEND = False                                                  # bench's instrumented step: per-launch timers need the launches to happen


class _Replay(Function):
    @staticmethod
    def forward(ctx, seg, entry, x):
        entry['x'].copy_(x)
        entry['fwd'].replay()
        seg._count_batchnorms()
        ctx.entry = entry
        outs = []
        for o in entry['outs']:
            d = o.detach()
            s16 = getattr(o, '_pdf_bf16', None)
            if s16 is not None:
                F.attach_shadow(d, s16)
            outs.append(d)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        e = ctx.entry
        for g, sg in zip(grads, e['gouts']):
            if g is None:
                sg.zero_()
            else:
                sg.copy_(g)
        main = torch.cuda.current_stream()
        side = F.wgrad_side_stream()
        for a, b in e['bwd']:                                  # last stage first
            a.replay()
            if b is not None:
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    b.replay()
        return None, None, e['gx'].detach()


class GraphedSegment:
    """stages: functions f_k with outs_k = f_k(outs_{k-1}) (outs_0 = f_0(x)), each a fixed chain of this library's Functions over
    modules[k] (whose parameters must all be Trainer-owned).  `segment(x)` -> (outs_0, .., outs_n) from graph replays; anything that
    does not qualify (no grad, eval mode, a capture already running, untagged parameters, switched off) calls the stages directly."""

    def __init__(self, stages, modules):
        self.stages = list(stages)
        self.modules = list(modules)
        self.entries = {}
        self.pool = None
        self.enabled = TRUNK_GRAPH == '1'
        self._bns = None
        self._params_ok = False

    def eager(self, x):
        outs = []
        for f in self.stages:
            x = f(x)
            outs.append(x)
        return tuple(outs)

    def _qualifies(self, x):
        if SUSPEND or not (self.enabled and x.is_cuda and torch.is_grad_enabled() and x.requires_grad) or torch.cuda.is_current_stream_capturing():
            return False
        if not all(m.training for m in self.modules):
            return False
        if not self._params_ok:                                # (checked until it holds once: the Trainer tags its parameters for good)
            for m in self.modules:
                for p in m.parameters():
                    if p.requires_grad and (not getattr(p, '_pdf_main_grad', False) or p.grad is None):
                        return False
            self._params_ok = True
        return True

    def _count_batchnorms(self):
        from .networks.layers import BatchNorm
        if self._bns is None:
            self._bns = [b for m in self.modules for b in m.modules() if isinstance(b, BatchNorm)]
        for b in self._bns:
            if b._pending == 0:
                BatchNorm._dirty.append(b)
            b._pending += 1

    def __call__(self, x):
        if not self._qualifies(x):
            return self.eager(x)
        key = (tuple(x.shape), x.dtype, tuple(x.stride()), F.gemm_precision(), F.storage_on(x.shape[0]))
        entry = self.entries.get(key)
        if entry is None:
            entry = self.entries[key] = self._capture(x)
        return _Replay.apply(self, entry, x)

    def _capture(self, x):
        from .networks.layers import BatchNorm
        if F._wg_used or F._wg_pending:
            # (weight-gradient work in flight belongs to a backward pass: the warm-up below would add to the same gradients)
            raise RuntimeError("pdfnet_amd: a graphed segment must be captured before the step's backward has started")
        params = [p for m in self.modules for p in m.parameters() if p.requires_grad]
        bufs = [b for m in self.modules for b in m.buffers()]
        BatchNorm.flush_counters()
        saved = [b.clone() for b in bufs]
        sx = x.detach().clone().requires_grad_(True)
        cur = torch.cuda.current_stream()
        s = torch.cuda.Stream()
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            for _ in range(WARMUP):
                outs = self.eager(sx)
                torch.autograd.grad(outs, [sx], [torch.ones_like(o) for o in outs])
                F.join_wgrad()
            del outs
        cur.wait_stream(s)
        BatchNorm.flush_counters()
        for b, v in zip(bufs, saved):
            b.copy_(v)
        if self.pool is None:
            self.pool = torch.cuda.graph_pool_handle()
        fwd = torch.cuda.CUDAGraph()
        with torch.cuda.graph(fwd, pool=self.pool):
            outs = self.eager(sx)
        gouts = [torch.zeros_like(o) for o in outs]
        ins = [sx] + list(outs[:-1])
        bwd, pins = [], []
        gin = None
        group_saved, F.WGRAD_GROUP = F.WGRAD_GROUP, 1 << 30     # defer every weight-gradient launch of the stage being captured
        try:
            for k in range(len(outs) - 1, -1, -1):
                a = torch.cuda.CUDAGraph()
                with torch.cuda.graph(a, pool=self.pool):
                    g = gouts[k] if gin is None else gouts[k] + gin     # this stage's output also feeds the next one
                    gin, = torch.autograd.grad([outs[k]], [ins[k]], [g])
                items = F.take_pending_wgrads()
                b = None
                if items:
                    b = torch.cuda.CUDAGraph()
                    # B_k runs beside A_{k-1} (captured later, same pool): what B_k reads and its workspaces must never be handed back
                    # to the pool while the graphs live, or A_{k-1}'s temporaries could be placed on top of them
                    F._ws_pins = ws_pins = []
                    try:
                        with torch.cuda.graph(b, pool=self.pool):
                            for it in items:
                                it[0]()
                    finally:
                        F._ws_pins = None
                    pins.append(([it[1] for it in items], ws_pins))
                pins.append((g, gin))
                bwd.append((a, b))
        finally:
            F.WGRAD_GROUP = group_saved
        # the capture passes executed nothing, but the Python side counted one BatchNorm call per layer; the first replay follows
        # immediately and counts this step's call itself
        for m in self.modules:
            for b_ in m.modules():
                if isinstance(b_, BatchNorm) and b_._pending > 0:
                    b_._pending -= 1
        BatchNorm._dirty = [b_ for b_ in BatchNorm._dirty if b_._pending > 0]
        for p in params:                                       # garbage from the warm-up passes
            p.grad.zero_()
        return {'x': sx, 'outs': tuple(outs), 'gouts': gouts, 'gx': gin, 'fwd': fwd, 'bwd': bwd, 'pins': pins}
