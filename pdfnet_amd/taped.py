"""A chain of sub-networks replayed from a TAPE of its library calls inside an otherwise eager train step.

Why (round 5): the step issues its ~1,700 launches from Python in ~31 ms of host time; the fp32 B=32 step needs 50 ms of GPU time, the bf16
B=32 step ~30 -- the latter is bound by the host (tools/host_time.py, profiles/r05_host_time.txt), and so is everything on a slow host.  Most of
those 31 ms are not launches but the Python around them: `autograd.Function.apply`, tensor allocation, option structures, attribute bookkeeping --
~35 us per Function call against ~4 us for the launch itself.  hipGraphs remove that time but replay slower than eager launches on this
runtime (profiles/r05_hipgraph_branches.txt, r05_trunk_graph_experiment.txt).  A tape removes it and keeps the launches exactly what they were:

  record   one pass through the segment (forward, then backward stage by stage) with every call into libpdfnet_hip.so -- function, integer /
           pointer / stream arguments, option structures -- appended to a list, and every tensor the pass allocates kept alive for good, so the
           recorded device addresses stay valid and nothing else is ever placed on top of them;
  replay   `for fn, args in tape: fn(*args)`: the same kernels with the same arguments on the same streams in the same order, including the
           fork / join events of the weight-gradient side stream -- only the Python in between is gone.

The ResNet trunk (layer1..layer4: 53 convolutions + 53 BatchNorms, a third of the step's launches, fixed shapes, no data-dependent control flow)
is the segment.  Same contract as torch.cuda.make_graphed_callables, restated for this library:
  * weight gradients are accumulated by the kernels straight into the trainer's flat gradient buffer (functional._main_grad), not returned to
    autograd -- only Trainer-owned parameters qualify; the recording's warm-up and recording passes (which add to those gradients and advance
    the BatchNorm running statistics) are undone, which is why recording happens in the forward of a step, before any real gradient exists;
  * `num_batches_tracked` is counted on the host (layers.BatchNorm._pending): a replay bumps the counters of the segment's BatchNorms itself;
  * bf16 shadows travel as Python attributes of tensors: the replayed outputs get theirs re-attached;
  * the few aten kernels inside the segment (gradient fan-in additions, zero fills) are taped as out-variant calls on the pinned tensors; any
    other aten kernel aborts the recording and the segment stays eager.
Results are bit-identical to the eager segment (`tests/test_trainer_gpu.py::test_taped_trunk_equals_the_eager_trunk_bit_for_bit`,
`::test_taped_trunk_inside_the_train_step`, `::test_taped_trunk_is_not_recorded_over_accumulated_gradients`).

Guard rails (round 6): a recording is REFUSED -- the call runs eagerly and the next qualifying call tries again -- while any gradient of the
segment's parameters is non-zero (the recording pass zeroes them afterwards: with a micro-batch already accumulated that would lose it);
`TapedSegment.pinned_bytes()` reports what the tapes hold alive (bench.py: config.taped_pinned_mb).
"""
import os

import torch
from torch.autograd import Function
from torch.utils._python_dispatch import TorchDispatchMode

from . import functional as F
from . import hip

TRUNK_TAPE = os.environ.get("PDFNET_TRUNK_TAPE", "1") != "0"
SUSPEND = False                                                  # bench's instrumented step: per-launch timers wrap the Python entry points

_ALLOC = ('empty', 'empty_like', 'empty_strided', 'new_empty', 'record_stream')
_VIEW = ('view', 'reshape', '_unsafe_view', 'permute', 'transpose', 'expand', 'slice', 'select', 'squeeze', 'unsqueeze', 't.', 'alias', 'detach',
         'as_strided', 'unbind', 'split', 'contiguous', '_reshape_alias', 'narrow', 'flatten', 'unflatten')


class _Recorder(TorchDispatchMode):
    """Pins every tensor the pass creates; tapes the aten KERNELS it meets (zero fills, additions) as calls on the pinned tensors."""

    def __init__(self, tape, pins):
        super().__init__()
        self.tape, self.pins, self.refused = tape, pins, []

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func).replace('aten.', '')
        self.pins.append(out)
        self.pins.append(args)                                   # (operands of taped aten calls and of library calls alike stay alive)
        if any(name.startswith(p) for p in _ALLOC) or any(name.startswith(p) for p in _VIEW):
            return out
        if name.startswith(('zeros_like', 'zeros.', 'new_zeros', 'zero_')):
            self.tape.append((out.zero_, ()))
        elif name.startswith('ones_like'):
            self.tape.append((out.fill_, (1.0,)))
        elif name == 'add.Tensor' and not kwargs and len(args) == 2 and all(torch.is_tensor(a) for a in args):
            a, b = args
            self.tape.append((lambda a=a, b=b, o=out: torch.add(a, b, out=o), ()))
        elif name.startswith('add_.Tensor') and len(args) == 2:
            a, b = args
            self.tape.append((lambda a=a, b=b: a.add_(b), ()))
        else:
            self.refused.append(name)
        return out


class _Replay(Function):
    @staticmethod
    def forward(ctx, seg, entry, x):
        entry['busy'] = True                                   # its activations are live until the backward has consumed them
        entry['x'].copy_(x)
        for fn, a in entry['fwd']:
            fn(*a)
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
        if hip._raw_stream(hip._raw_device()) != e['stream']:
            raise RuntimeError("pdfnet_amd: a taped segment was recorded on another stream than the one its backward runs on")
        for g, sg in zip(grads, e['gouts']):
            if g is None:
                sg.zero_()
            else:
                sg.copy_(g)
        for tape in e['bwd']:                                  # last stage first
            for fn, a in tape:
                fn(*a)
        F._wg_used.update(e['wg_keys'])                        # the taped weight-gradient launches sit on these side streams: join_wgrad waits
        e['busy'] = False
        return None, None, e['gx'].detach()


class TapedSegment:
    """stages: functions f_k with outs_k = f_k(outs_{k-1}) (outs_0 = f_0(x)), each a fixed chain of this library's Functions over
    modules[k] (whose parameters must all be Trainer-owned).  `segment(x)` -> (outs_0, .., outs_n) from tape replays; anything that
    does not qualify (no grad, eval mode, a stream capture running, untagged parameters, switched off) calls the stages directly."""

    MAX_ENTRIES = 2

    def __init__(self, stages, modules):
        self.stages = list(stages)
        self.modules = list(modules)
        self.entries = {}
        self.enabled = TRUNK_TAPE
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
        if hip._raw_stream(hip._raw_device()) != 0:            # the step is issued on the default stream; warm-up passes on other streams stay eager
            return False
        if not all(m.training for m in self.modules) or F.WGRAD_GROUP > 1 or not (F.ASYNC_WGRAD and F.USE_SIDE_STREAMS):
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
        key = (tuple(x.shape), x.dtype, tuple(x.stride()), F._GEMM_BF16, F.storage_on(x.shape[0]), hip._raw_stream(hip._raw_device()))
        entry = self.entries.get(key)
        if entry is None:
            if len(self.entries) >= self.MAX_ENTRIES:          # every signature pins its own activations: a stream of changing shapes stays eager
                return self.eager(x)
            entry = self._record(x)
            if entry is None:                                  # gradients already hold something: not now (see _record)
                return self.eager(x)
            self.entries[key] = entry
        if entry is False or entry['busy']:                    # refused once for this signature / a second forward before the first one's backward
            return self.eager(x)
        return _Replay.apply(self, entry, x)

    def _record(self, x):
        from .networks.layers import BatchNorm
        if F._wg_used or F._wg_pending:
            # (weight-gradient work in flight belongs to a backward pass: the passes below would add to the same gradients)
            raise RuntimeError("pdfnet_amd: a taped segment must be recorded before the step's backward has started")
        params = [p for m in self.modules for p in m.parameters() if p.requires_grad]
        # the recording pass ends by zeroing these gradients (it is a real pass that must leave no trace): refuse while they hold a previous
        # micro-batch of an accumulation loop -- one reduction and one host read, once per signature
        if params and float(torch.stack(torch._foreach_norm([p.grad for p in params])).sum()) != 0.0:
            return None
        bufs = [b for m in self.modules for b in m.buffers()]
        BatchNorm.flush_counters()
        saved = [b.clone() for b in bufs]
        pins, fwd, bwd = [], [], []
        sx = x.detach().clone().requires_grad_(True)
        multi = torch.autograd.is_multithreading_enabled()
        torch.autograd.set_multithreading_enabled(False)       # the backward must run on THIS thread: the recorder is thread-local
        rec = _Recorder(fwd, pins)
        try:
            with rec:
                hip._tape = fwd
                outs = self.eager(sx)
                hip._tape = None
                gouts = [torch.zeros_like(o) for o in outs]
                del fwd[len(fwd) - len(outs):]                 # (the zero fills of gouts just taped: they are inputs, filled by the replay)
                ins = [sx] + list(outs[:-1])
                gin = None
                wg_before = set(F._wg_used)
                for k in range(len(outs) - 1, -1, -1):
                    tape = []
                    rec.tape = tape
                    hip._tape = tape
                    g = gouts[k] if gin is None else gouts[k] + gin     # this stage's output also feeds the next one
                    gin, = torch.autograd.grad([outs[k]], [ins[k]], [g])
                    hip._tape = None
                    bwd.append(tape)
                    pins.append((g, gin))
        finally:
            hip._tape = None
            torch.autograd.set_multithreading_enabled(multi)
        wg_keys = set(F._wg_used) - wg_before
        F.join_wgrad()                                         # the recording pass was a real pass: its weight gradients are in flight
        # undo what the recording pass did to persistent state: running statistics, counters, gradients
        BatchNorm.flush_counters()
        for b, v in zip(bufs, saved):
            b.copy_(v)
        for p in params:
            p.grad.zero_()
        if rec.refused:
            import warnings
            warnings.warn("pdfnet_amd: segment not taped, it runs aten kernels the tape does not know: %s" % sorted(set(rec.refused)))
            return False
        return {'busy': False, 'x': sx, 'outs': tuple(outs), 'gouts': gouts, 'gx': gin, 'fwd': fwd, 'bwd': bwd, 'pins': pins, 'wg_keys': wg_keys,
                'stream': hip._raw_stream(hip._raw_device()), 'pinned_bytes': _storage_bytes(pins)}

    def pinned_bytes(self):
        """Device bytes the recorded tapes keep alive for the life of the process (every activation of every recorded signature)."""
        return sum(e['pinned_bytes'] for e in self.entries.values() if e)


def _storage_bytes(obj, seen=None):
    """Bytes of the distinct device storages reachable from a nest of tuples / lists of tensors."""
    seen = {} if seen is None else seen
    stack = [obj]
    while stack:
        o = stack.pop()
        if torch.is_tensor(o):
            if o.is_cuda:
                s = o.untyped_storage()
                seen[s.data_ptr()] = max(seen.get(s.data_ptr(), 0), s.nbytes())
        elif isinstance(o, (tuple, list)):
            stack.extend(o)
        elif isinstance(o, dict):
            stack.extend(o.values())
    return sum(seen.values())
