"""A sub-network replayed as two hipGraphs (forward, backward) inside an otherwise eager train step.

Why (round 5): the step issues ~1,800 launches from Python in ~31 ms of host time; at fp32 B=32 the GPU needs 51 ms and the host
is never the limit, at bf16 B=32 the GPU needs ~30 ms and the host IS the limit (tools/host_time.py).  The whole step as ONE hipGraph
costs no host time but replays slower than eager streams: every cross-stream edge inside a captured graph costs ~14 us of the main
chain on this runtime (profiles/r05_hipgraph_branches.txt) and the step has hundreds.  The ResNet trunk is the part of the step
that is a plain chain -- 53 convolutions + 53 BatchNorms, a third of all launches, no data-dependent control flow, fixed shapes --
so it alone is captured: its forward is a linear graph (one submission: ~10 us of host time for ~350 kernels), its backward a chain
plus the weight-gradient side branch.  Everything around it stays eager and keeps its stream overlap.

The mechanism is the one of torch.cuda.make_graphed_callables, restated for this library's conventions:
  * weight gradients are accumulated by the backward kernels straight into the trainer's flat gradient buffer (functional._main_grad), not
    returned to autograd -- so only Trainer-owned parameters qualify, and the capture's warm-up passes (which add garbage there) are
    followed by re-zeroing those gradients; call sites must capture before any real gradient is in the buffer (the forward of a step);
  * BatchNorm running statistics advance in the kernels (replayed) but `num_batches_tracked` is counted on the host
    (layers.BatchNorm._pending): the replay bumps the counters of the segment's BatchNorms itself; the warm-up's extra updates of
    the running statistics are rolled back;
  * bf16 shadows travel as Python attributes of tensors (functional.attach_shadow): the replayed outputs get theirs re-attached.
"""
import os

import torch
from torch.autograd import Function

from . import functional as F

TRUNK_GRAPH = os.environ.get("PDFNET_TRUNK_GRAPH", "0")         # '0' eager (default), '1' graphed under a Trainer
BWD_WGRAD_GROUP = int(os.environ.get("PDFNET_TRUNK_GRAPH_WGRAD_GROUP", "1"))   # weight gradients per side-stream edge inside the backward graph
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
        e['bwd'].replay()
        return None, None, e['gx'].detach()


class GraphedSegment:
    """fn(x) -> tuple of tensors, a fixed chain of this library's Functions over `modules` (their parameters must all be Trainer-owned).
    `segment(x)` replays it; anything that does not qualify (no grad, eval mode, a capture already running, untagged parameters,
    switched off) calls `fn` directly."""

    def __init__(self, fn, modules):
        self.fn = fn
        self.modules = list(modules)
        self.entries = {}
        self.pool = None
        self.enabled = TRUNK_GRAPH == '1'
        self._bns = None
        self._params_ok = False

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
            return self.fn(x)
        key = (tuple(x.shape), x.dtype, tuple(x.stride()), F.gemm_precision(), F.storage_on(x.shape[0]))
        entry = self.entries.get(key)
        if entry is None:
            entry = self.entries[key] = self._capture(x)
        return _Replay.apply(self, entry, x)

    def _capture(self, x):
        from .networks.layers import BatchNorm
        if F._wg_used or F._wg_pending:
            # (weight-gradient side streams with eager work in flight would be joined INTO the capture by join_wgrad below)
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
                outs = self.fn(sx)
                torch.autograd.grad(outs, [sx], [torch.ones_like(o) for o in outs])
                F.join_wgrad()
            del outs
        cur.wait_stream(s)
        BatchNorm.flush_counters()
        for b, v in zip(bufs, saved):
            b.copy_(v)
        if self.pool is None:
            self.pool = torch.cuda.graph_pool_handle()
        fwd, bwd = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(fwd, pool=self.pool):
            outs = self.fn(sx)
        outs = tuple(outs)
        gouts = [torch.zeros_like(o) for o in outs]
        group_saved, F.WGRAD_GROUP = F.WGRAD_GROUP, max(F.WGRAD_GROUP, BWD_WGRAD_GROUP)
        try:
            with torch.cuda.graph(bwd, pool=self.pool):
                gx, = torch.autograd.grad(outs, [sx], gouts)
                F.join_wgrad()                                 # the weight-gradient branch joins the chain inside the graph
        finally:
            F.WGRAD_GROUP = group_saved
        # the capture passes themselves executed nothing, but the Python side counted one BatchNorm call per layer: that is THIS step's
        # count (the first replay follows immediately and counts again) -- take the capture's count back
        for m in self.modules:
            for b in m.modules():
                if isinstance(b, BatchNorm) and b._pending > 0:
                    b._pending -= 1
        BatchNorm._dirty = [b for b in BatchNorm._dirty if b._pending > 0]
        for p in params:                                       # garbage from the warm-up passes
            p.grad.zero_()
        return {'x': sx, 'outs': outs, 'gouts': gouts, 'gx': gx, 'fwd': fwd, 'bwd': bwd}
