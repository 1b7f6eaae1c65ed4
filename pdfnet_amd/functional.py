"""Autograd shells around the C-ABI kernels (libpdfnet_hip.so).  PyTorch here is plumbing only:
device memory, the current stream and the autograd tape; every computation below is a HIP kernel.

Activation layout: 4-D maps are logical NCHW tensors in torch.channels_last memory (physically NHWC,
rows = pixels, channels contiguous); everything else is row-major [..., channels].
"""
import ctypes
import torch
from torch.autograd import Function

from . import hip
from .hip import ptr, ptr_at, stream

ACT_NONE, ACT_RELU, ACT_LRELU = 0, 1, 2
import os as _os

_byref = ctypes.byref
_CallOpts = hip.CallOpts
USE_SIDE_STREAMS = _os.environ.get("PDFNET_SIDE_STREAMS", "1") != "0"
_side = {}


def _record(obj, stream_):
    if torch.is_tensor(obj):
        obj.record_stream(stream_)
        s16 = getattr(obj, '_pdf_bf16', None)           # its bf16 shadow travels with it
        if s16 is not None:
            s16.record_stream(stream_)
    elif isinstance(obj, (list, tuple)):
        for o in obj:
            _record(o, stream_)
    elif isinstance(obj, dict):
        for o in obj.values():
            _record(o, stream_)


class fork:
    """Start `fn` on a side HIP stream now; `.join()` makes the current stream wait for it and returns its result.
    Lets independent sub-networks (sibling decoders, heads) overlap the main chain instead of queueing behind it."""
    _busy = {}

    def __init__(self, fn):
        self.stream = None
        if not USE_SIDE_STREAMS:
            self.out = fn()
            return
        cur = torch.cuda.current_stream()
        dev = torch.cuda.current_device()
        pool = _side.setdefault(('fork', dev), [])
        busy = fork._busy.setdefault(dev, set())
        free = [i for i in range(len(pool)) if i not in busy]
        if not free:
            pool.append(torch.cuda.Stream())
            free = [len(pool) - 1]
        self.idx, self.dev = free[0], dev
        busy.add(self.idx)
        self.stream = pool[self.idx]
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            self.out = fn()

    def join(self):
        if self.stream is not None:
            cur = torch.cuda.current_stream()
            cur.wait_stream(self.stream)
            _record(self.out, cur)
            fork._busy[self.dev].discard(self.idx)
            self.stream = None
        return self.out


def parallel(*fns):
    """Run independent branches (left/right hand sub-networks, sibling decoders) on separate HIP streams.
    Their kernels are launch-bound and fill a fraction of the 256 CUs each; forked streams let them overlap, and
    inside a captured step the fork/join becomes parallel branches of the hipGraph.  Autograd replays every
    node's backward on the stream its forward ran on, so the backward overlaps the same way."""
    if not USE_SIDE_STREAMS or len(fns) == 1:
        return [f() for f in fns]
    cur = torch.cuda.current_stream()
    key = (cur.device_index if hasattr(cur, 'device_index') else torch.cuda.current_device())
    pool = _side.setdefault(key, [])
    while len(pool) < len(fns) - 1:
        pool.append(torch.cuda.Stream())
    streams = [cur] + pool[:len(fns) - 1]
    for s_ in streams[1:]:
        s_.wait_stream(cur)
    out = []
    for f, s_ in zip(fns, streams):
        with torch.cuda.stream(s_):
            out.append(f())
    for r, s_ in zip(out[1:], streams[1:]):
        cur.wait_stream(s_)
        _record(r, cur)                     # produced on a side stream, consumed on the main one
    return out

CL = torch.channels_last


def set_gemm_precision(dtype):
    """'fp32' (default): every contraction on the exact fp32-input MFMA.  'bf16': operands of the conv / transposed-conv /
    linear kernels (forward, backward-data, weight gradient) are rounded to bf16 as they are staged, bf16 MFMA with fp32
    accumulation; weights, activations, normalisation statistics, losses and the optimizer stay fp32 (BASELINE configs 4-5)."""
    if dtype not in ('fp32', 'bf16'):
        raise ValueError("pdfnet_amd: gemm precision must be 'fp32' or 'bf16'")
    global _GEMM_BF16
    hip.lib().pdf_set_gemm_precision(1 if dtype == 'bf16' else 0)
    _GEMM_BF16 = dtype == 'bf16'                          # cached for the per-launch checks (shadows_on)


def gemm_precision():
    return 'bf16' if hip.lib().pdf_debug_gemm_precision() else 'fp32'


# ---- bf16 shadows (bf16 mode): BatchNorm (forward output, backward dx), the pyramid L2Norm and the trainer (weights) write a bf16
# copy of what they produce; the conv / deconv / linear launches that read those tensors hand the copy to the library
# (PdfCallOpts::op0_bf16 / op1_bf16), whose bf16 GEMM kernels then stage 2-byte operands.  Results are bit-identical to the plain bf16 mode
# (same round-to-nearest-even, done by the producer instead of the consumer).  A shadow travels as an attribute of the tensor
# it mirrors and is ignored once that tensor has been modified in place.
BF16_SHADOWS = _os.environ.get("PDFNET_BF16_SHADOWS", "1") != "0"


_GEMM_BF16 = False


def shadows_on():
    return BF16_SHADOWS and _GEMM_BF16


def attach_shadow(t, s16):
    t._pdf_bf16, t._pdf_bf16_ver = s16, t._version
    return t


def shadow_of(t):
    """The bf16 copy of `t` if it has a valid one with the same layout, else None."""
    if t is None or not (BF16_SHADOWS and _GEMM_BF16):
        return None
    s16 = getattr(t, '_pdf_bf16', None)
    if s16 is None or t._version != getattr(t, '_pdf_bf16_ver', -1) or s16.shape != t.shape or s16.stride() != t.stride():
        return None
    return s16


# Transposed bf16 shadows of the weights (trains.base_trainer.FlatAdam._cast_transposed): valid exactly when the plain shadow is.
TRANSPOSED_SHADOWS = _os.environ.get("PDFNET_BF16_TRANSPOSED", "1") != "0"


def shadow_t_of(w):
    if not TRANSPOSED_SHADOWS or shadow_of(w) is None:
        return None
    return getattr(w, '_pdf_bf16_t', None)


def new_shadow(t):
    return torch.empty_like(t, dtype=torch.bfloat16) if shadows_on() else None


# ---- bf16 STORAGE of the conv -> BatchNorm tensors (bf16 mode, PDFNET_BF16_STORAGE).  A convolution called with stats=True
# feeds nothing but a training-mode BatchNorm.  Its output then exists only as bf16 (written by the GEMM epilogue, read by the
# BatchNorm kernels: 2 instead of 4 bytes in five passes), and so does the gradient the BatchNorm hands back (it already wrote a
# bf16 shadow of it for the backward GEMMs; now it writes nothing else).  Autograd still sees fp32 tensors of the right shape --
# allocated, never written ("phantoms"), carrying the bf16 tensor as an attribute; the library gets a NULL fp32 pointer for them
# and refuses (PDF_E_BADARG) any launch that would have to read it.
# Default 'auto': on for convolutions over >= BF16_STORAGE_MIN_BATCH images (48 until round 5, when the B=32 step stopped being bound by the host:
# storage + epilogue statistics 954 -> 990 img/s there, profiles/r05_bf16_b32_modes.txt; now 32) -- measured per GPU: B=64 1,046-1,050 -> 1,063-1,076 img/s (the
# covered BatchNorm sites 9.5 -> 5.5 ms), while at B=32 the step is bound by the host's issue time and the extra allocations and calls
# cost more than the kernels gain (868 -> 799).  PDFNET_BF16_STORAGE=1 / 0 forces it on / off.
_bs = _os.environ.get("PDFNET_BF16_STORAGE", "auto")
BF16_STORAGE = 'auto' if _bs == "auto" else _bs != "0"
BF16_STORAGE_MIN_BATCH = int(_os.environ.get("PDFNET_BF16_STORAGE_MIN_BATCH", "32"))


def storage_on(batch=None):
    """bf16 storage for a convolution over `batch` images (None: only when forced on)."""
    if not (BF16_SHADOWS and _GEMM_BF16):
        return False
    if BF16_STORAGE == 'auto':
        return batch is not None and batch >= BF16_STORAGE_MIN_BATCH
    return bool(BF16_STORAGE)


def _lazy_bn(t):
    """(raw tensor, scale, shift) if `t` is the never-written output of a lazy BatchNorm + ReLU (see _BatchNorm, lazy=True)."""
    v = getattr(t, '_pdf_lazy', None)
    if v is None or v[3] != t._version:
        return None
    return v[:3]


# debug aid for the bf16 storage mode: phantoms (allocated, never written) are NaN-filled so that a stray reader cannot go unnoticed
DEBUG_PHANTOMS = _os.environ.get("PDFNET_DEBUG_PHANTOMS", "0") != "0"


def _stored16(t):
    """The bf16 tensor that IS the value of phantom `t` (None for an ordinary tensor)."""
    v = getattr(t, '_pdf_y16', None)
    if v is None:
        return None
    if v[1] != t._version:
        # a phantom's fp32 storage was never written: once it has been modified in place there is no value left to read
        raise RuntimeError("pdfnet_amd: a bf16-storage phantom tensor was modified in place; its fp32 storage holds no value (PDFNET_BF16_STORAGE)")
    return v[0]



def _O(**kw):
    """The explicit per-call options of an `_x` entry point (include/pdfnet_hip.h PdfCallOpts): -> (structure, argument).  Every field a
    call takes beyond its positional arguments -- bf16 shadows of its operands, a bf16 output, a statistics request, an operand
    transform -- is in ITS argument list; nothing is armed on the thread for "the next call"."""
    o = None
    for k, v in kw.items():                                  # (the common case -- fp32 mode, no statistics -- allocates nothing)
        if v is not None:
            if o is None:
                o = _CallOpts()
            setattr(o, k, v)
    if o is None:
        return None, None
    return o, _byref(o)


# Winograd (csrc/winograd.hip, fp32 mode): a stride-1 3x3 convolution with >= 128 channels is handed a workspace and the library computes it
# in the transform domain -- F(4x4, 3x3) (4x fewer MFMA instructions, ~4e-5 absolute error) where the map and PDF_WINOGRAD_F4 allow, else
# F(2x2, 3x3) (2.25x fewer, more accurate than the direct sum); forward, backward-data and the weight gradient.  The
# workspace is a fresh torch.empty per call (GB-scale for `feat`: the caching allocator keeps a few such blocks per stream).
WINOGRAD = _os.environ.get("PDFNET_WINOGRAD", "1") != "0"
WINOGRAD_INFERENCE = _os.environ.get("PDFNET_WINOGRAD_INFERENCE", "0") != "0"     # also for forwards without autograd (eval / no_grad)
_wino_cache = {}


def _wino_ws(N, H, W, Cin, Cout, KH, KW, stride, pad, backward, dev):
    """-> (workspace tensor, floats) for the Winograd path of this convolution's forward (backward = 0) or backward-data pass, or (None, None)."""
    if not WINOGRAD or _GEMM_BF16 or KH != 3 or KW != 3 or stride != 1 or pad != 1 or Cin < 64 or Cout < 64:
        return None, None
    key = (N, H, W, Cin, Cout, backward)
    n = _wino_cache.get(key)
    if n is None:
        n = _wino_cache[key] = _L().pdf_conv2d_winograd_workspace_floats(N, H, W, Cin, Cout, KH, KW, stride, pad, backward)
    if n <= 0:
        return None, None
    return torch.empty(n, dtype=torch.float32, device=dev), n


# x3 arithmetic for the pyramid's kernel == stride transposed convolutions (csrc/gemm_x3.hip: fp32 products as six bf16 MFMAs on 3-way split
# operands, fp32-exact to ~2^-24): the library asks for a workspace for the split operands; 0 floats = the layer does not qualify.
X3_DECONV = _os.environ.get("PDFNET_X3_DECONV", "1") != "0"
_x3d_cache = {}


def set_x3(mode):
    """Which launches run as x3 products (include/pdfnet_hip.h pdf_set_x3_mode: bit 0 Winograd-domain products of the wide layers, bit 1 the
    transposed convolutions, bit 2 the fused mesh decoder's linear products; None = the environment's choice).  Drops the cached workspace sizes, which
    depend on it."""
    _L().pdf_set_x3_mode(-1 if mode is None else int(mode))
    _wino_cache.clear()
    _wino_voff.clear()
    _x3d_cache.clear()


def _x3_deconv_ws(N, H, W, Cin, Cout, KH, KW, stride, pad, backward, dev):
    if not X3_DECONV or _GEMM_BF16 or KH != KW or stride < 2:
        return None, None
    key = (N, H, W, Cin, Cout, KH, stride, pad, backward)
    n = _x3d_cache.get(key)
    if n is None:
        n = _x3d_cache[key] = _L().pdf_deconv2d_x3_workspace_floats(N, H, W, Cin, Cout, KH, KW, stride, pad, backward)
    if n <= 0:
        return None, None
    return torch.empty(n, dtype=torch.float32, device=dev), n


WINOGRAD_KEEP_V = _os.environ.get("PDFNET_WINOGRAD_KEEP_V", "1") != "0"
_wino_voff = {}


def _wino_v_offset(N, H, W, Cin, Cout, KH, KW, stride, pad):
    """Float offset of the transformed input V inside this convolution's forward Winograd workspace, or -1 (no F(4x4) forward, or the
    weight gradient does not take the Winograd path)."""
    key = (N, H, W, Cin, Cout, KH, KW, stride, pad)
    off = _wino_voff.get(key)
    if off is None:
        off = _wino_voff[key] = _L().pdf_conv2d_winograd_v_offset(N, H, W, Cin, Cout, KH, KW, stride, pad)
    return off


WINOGRAD_SHARE = _os.environ.get("PDFNET_WINOGRAD_SHARE", "1") != "0"


def share_winograd_input(x):
    """Mark `x` as read by several stride-1 3x3 convolutions: their F(4x4) forwards (and weight gradients) share one transformed input."""
    if WINOGRAD_SHARE and x is not None and getattr(x, '_pdf_wino_share', None) is None:
        x._pdf_wino_share = {}
    return x


def _O2(a, b):
    """_O2(a, b) for the most frequent call shape: the two operand shadows of a GEMM-family launch."""
    if a is None and b is None:
        return None
    o = _CallOpts()
    if a is not None:
        o.op0_bf16 = a.data_ptr()
    if b is not None:
        o.op1_bf16 = b.data_ptr()
    return _byref(o)


# ---- BatchNorm statistics out of the producing GEMM's epilogue (fp32 and bf16 kernels): a conv / linear forward called with stats=True
# asks the library for per-row-block (mean, M2) pairs of its output columns (PdfCallOpts::stats_out); they travel as an attribute
# of the output tensor and the BatchNorm that consumes it skips its own statistics pass over the tensor (PdfCallOpts::tile_stats).
BN_EPILOGUE_STATS = _os.environ.get("PDFNET_BN_EPILOGUE_STATS", "1") != "0"
# bf16 mode: the kernels can do it too (whole tiles).  Round 3 (register-staged bf16 kernels): B=32 850 -> 868 img/s, B=64 1,038 -> 1,013, so it
# was opt-in.  Round 4 (LDS-DMA kernels): B=64 1,098 -> 1,120, B=32 (bound by the host's issue time) 818 -> 717 on a slow host
# (profiles/r04_bf16_storage_ab.txt).  Default 'auto': convolutions over >= BF16_STORAGE_MIN_BATCH images, like the bf16 storage; 1 / 0 force it.
_be = _os.environ.get("PDFNET_BN_EPILOGUE_STATS_BF16", "auto")
BN_EPILOGUE_STATS_BF16 = 'auto' if _be == "auto" else _be != "0"


def bf16_modes(batch):
    """What the bf16 mode resolves to for convolutions over `batch` images -- the 'auto' rules switch at BF16_STORAGE_MIN_BATCH, so a
    B=32 and a B=64 run of the same model follow different rounding models; the Trainer logs this once per run."""
    if not _GEMM_BF16:
        return {'gemm': 'fp32'}
    stats = (batch >= BF16_STORAGE_MIN_BATCH) if BN_EPILOGUE_STATS_BF16 == 'auto' else bool(BN_EPILOGUE_STATS_BF16)
    return {'gemm': 'bf16', 'shadows': BF16_SHADOWS, 'conv_to_bn_storage': 'bf16' if storage_on(batch) else 'fp32',
            'bn_statistics': ('fp32 accumulators in the GEMM epilogue' if (stats and BN_EPILOGUE_STATS) else "the BatchNorm's own pass over the stored tensor"),
            'auto_threshold_batch': BF16_STORAGE_MIN_BATCH}


def _stats_request(stats, rows, cols, dev, batch=None):
    """-> the partials buffer for a conv / linear forward launch (PdfCallOpts::stats_out, cap = numel), or None.
    batch: images the launch covers (convolutions), for the bf16 'auto' rule."""
    if not (stats and BN_EPILOGUE_STATS):
        return None
    if _GEMM_BF16:
        on = (batch is not None and batch >= BF16_STORAGE_MIN_BATCH) if BN_EPILOGUE_STATS_BF16 == 'auto' else bool(BN_EPILOGUE_STATS_BF16)
        if not on:
            return None
    cap = ((rows + 31) // 32) * cols * 2
    return torch.empty(cap, dtype=torch.float32, device=dev)


def _stats_attach(y, part, o):
    """o: the call's PdfCallOpts after the call (stats_tiles / stats_rows written back by the library)."""
    if part is None:
        return
    if o.stats_tiles > 0:
        y._pdf_bn_tiles = (part, o.stats_tiles, o.stats_rows, y._version)


def tile_stats_of(x):
    t = getattr(x, '_pdf_bn_tiles', None)
    if t is None or t[3] != x._version:
        return None
    return t


def carry_stats(src, dst):
    """`dst` is a re-shaped VIEW of `src` with the same rows x channels: it keeps the statistics partials."""
    t = getattr(src, '_pdf_bn_tiles', None)
    if t is not None and dst.data_ptr() == src.data_ptr():
        dst._pdf_bn_tiles = t
    return dst


class _forced_fp32:
    """Launches issued inside run on the fp32 kernels even in bf16 mode (the precision flag is read on the host at launch)."""

    def __init__(self, on):
        self.on = on and hip.lib().pdf_debug_gemm_precision() != 0

    def __enter__(self):
        if self.on:
            hip.lib().pdf_set_gemm_precision(0)

    def __exit__(self, *exc):
        if self.on:
            hip.lib().pdf_set_gemm_precision(1)
        return False


def _L():
    return hip.lib()


def cl(x):
    """4-D tensor -> channels_last-contiguous (no copy if it already is)."""
    return x.contiguous(memory_format=CL)


def _rows(x):
    """(R, C) of a row-major / channels_last tensor."""
    if x.dim() == 4:
        return x.shape[0] * x.shape[2] * x.shape[3], x.shape[1]
    return x.numel() // x.shape[-1], x.shape[-1]


def _canon(x):
    return cl(x) if x.dim() == 4 else x.contiguous()


def _zeros_cl(shape, dev):
    return torch.empty(shape, dtype=torch.float32, device=dev, memory_format=CL).zero_()


def _ws(nfloats, dev):
    return torch.empty(max(int(nfloats), 1), dtype=torch.float32, device=dev)


_wgrad_ws_cache = {}
_bn_ws_cache = {}


def _wgrad_ws(M, NI, NJ, dev):
    n = _wgrad_ws_cache.get((M, NI, NJ))                    # (the sizes depend on the shape only: one library call per shape, not per launch)
    if n is None:
        n = _wgrad_ws_cache[(M, NI, NJ)] = _L().pdf_wgrad_workspace_floats(M, NI, NJ)
    return _ws(n, dev), n


def _bn_ws_floats(C, R):
    n = _bn_ws_cache.get((C, R))
    if n is None:
        n = _bn_ws_cache[(C, R)] = _L().pdf_bn_workspace_floats(C, R)
    return n


ASYNC_WGRAD = _os.environ.get("PDFNET_ASYNC_WGRAD", "1") != "0"
_wg_streams = {}
_wg_used = set()


# queue priority of the weight-gradient side streams (torch: 0 = normal, -1 = high; the default stream the step is issued on outranks a
# normal-priority stream in practice)
WGRAD_STREAM_PRIORITY = int(_os.environ.get("PDFNET_WGRAD_STREAM_PRIORITY", "0"))


class wgrad_stream:
    """Context for weight-gradient kernels that accumulate straight into the trainer's flat gradient buffer: nothing
    in the backward chain consumes them, so they run on a side HIP stream and overlap the data-gradient chain (whose
    small layers fill only part of the 256 CUs).  `join_wgrad()` is called once after backward.

    There is one side stream per launching stream, so two branches' weight gradients overlap each other too.  A parameter
    that is used by modules running on DIFFERENT streams (a shared layer called from two forked branches) would get two
    non-atomic `dW += ...` launches on two side streams at once: `params` names the accumulation targets, and a side
    stream first waits for the stream that last accumulated into the same parameter."""

    def __init__(self, enabled, *tensors, params=(), also_wait=None):
        self.enabled = enabled and ASYNC_WGRAD and USE_SIDE_STREAMS
        self.tensors = tensors
        self.params = params
        self.also_wait = also_wait                  # raw stream the inputs were produced on, when it is not the current one (flush_wgrad)

    def __enter__(self):
        if not self.enabled:
            return self
        # raw handles + the library's event ring: the torch.cuda.stream() / Stream.wait_stream() route builds four Python
        # Stream / Event objects per switch through the device-index helpers (~35 us, x180 per step = a sixth of the
        # backward's host time)
        dev = hip._raw_device()
        cur_raw = hip._raw_stream(dev)
        key = (dev, cur_raw)
        ent = _wg_streams.get(key)
        if ent is None:
            if WGRAD_STREAM_PRIORITY > 0:                    # below normal: only the HIP API offers it (pdf_stream_create)
                import ctypes as _ct
                raw = _ct.c_void_p()
                _L().pdf_stream_create(_ct.byref(raw), WGRAD_STREAM_PRIORITY, None, None)
                s = torch.cuda.ExternalStream(raw.value)
            else:
                s = torch.cuda.Stream(priority=WGRAD_STREAM_PRIORITY)
            ent = _wg_streams[key] = (s, s.cuda_stream, s.stream_id, s.device_index, s.device_type)
        side, side_raw = ent[0], ent[1]
        wait = _L().pdf_stream_wait
        wait(side_raw, cur_raw)
        if self.also_wait is not None and self.also_wait != cur_raw:
            wait(side_raw, self.also_wait)
        for p in self.params:
            if p is not None:
                last = getattr(p, '_pdf_wg_last', None)
                if last is not None and last is not ent:
                    wait(side_raw, last[1])
                p._pdf_wg_last = ent
        for t in self.tensors:                      # produced / owned by the main stream, read on the side stream
            if t is not None:                       # (keeping them referenced until the join instead was measured: no difference, r05_keepalive.txt)
                t.record_stream(side)
                s16 = getattr(t, '_pdf_bf16', None)
                if s16 is not None:
                    s16.record_stream(side)
        _wg_used.add(key)
        self._prev = torch._C._cuda_getCurrentStream(dev)
        torch._C._cuda_setStream(stream_id=ent[2], device_index=ent[3], device_type=ent[4])
        return self

    def __exit__(self, *exc):
        if self.enabled:
            p = self._prev
            torch._C._cuda_setStream(stream_id=p[0], device_index=p[1], device_type=p[2])
        return False


# Weight gradients in groups (round 5).  Every switch to the side stream is one cross-stream edge: an event record in the main chain
# (eager: ~3 us of queue bubble per edge; inside a captured hipGraph ~14 us per edge -- profiles/r05_hipgraph_branches.txt, the reason the
# forked capture replayed slower than the single-stream one) plus ~11 us of host time.  With WGRAD_GROUP = G > 1 the launches are kept as
# closures (which hold their operands alive) and issued G at a time behind ONE edge.  Nothing reads a weight gradient before join_wgrad /
# the trainer's early-slice hook, and both flush first.  Lists are per launching stream: a group is flushed while that stream is current.
WGRAD_GROUP = int(_os.environ.get("PDFNET_WGRAD_GROUP", "1"))
_wg_pending = {}


def _wg_defer(fn, tensors, params):
    key = hip._raw_stream(hip._raw_device())
    lst = _wg_pending.setdefault(key, [])
    lst.append((fn, tensors, params))
    if len(lst) >= WGRAD_GROUP:
        _wg_flush_key(key)


def _wg_flush_key(key):
    items = _wg_pending.pop(key, None)
    if not items:
        return
    with wgrad_stream(True, *[t for it in items for t in it[1]], params=[p for it in items for p in it[2]], also_wait=key):
        for it in items:
            it[0]()


def flush_wgrad():
    """Issue every deferred weight-gradient launch now (on the current stream's side stream, after the stream each was deferred on)."""
    for key in list(_wg_pending):
        _wg_flush_key(key)


def join_wgrad(keep=False):
    """Make the current stream wait for every outstanding side-stream weight-gradient kernel.  keep: the streams stay listed
    (a later join -- another stream's, or the one after the backward -- still waits for them)."""
    flush_wgrad()
    cur = hip.stream()
    for key in list(_wg_used):
        _L().pdf_stream_wait(cur, _wg_streams[key][1])
    if not keep:
        _wg_used.clear()


def _main_grad(param, like):
    """The trainer's flat gradient view of `param` if gradients may be accumulated into it directly
    (FlatAdam tags its parameters): skips autograd's separate `grad += dw` pass and the dw allocation."""
    if param is None or not getattr(param, '_pdf_main_grad', False):
        return None
    g = getattr(param, '_pdf_grad_alias', None)            # a re-shaped view of a parameter carries the matching view of its gradient
    if g is None:
        g = param.grad
    if g is None or g.shape != like.shape or g.stride() != like.stride():
        return None
    return g


def _colsum_into(g, C, R, ldg, out):
    ws = _ws(_bn_ws_floats(C, R), g.device)
    _L().pdf_colsum(ptr(g), ldg, C, R, ptr(out), 1, ptr(ws), stream())


def _colsum(g, C, R, ldg):
    out = torch.empty(C, dtype=torch.float32, device=g.device)
    ws = _ws(_bn_ws_floats(C, R), g.device)
    _L().pdf_colsum(ptr(g), ldg, C, R, ptr(out), 0, ptr(ws), stream())
    return out


ASYNC_WGRAD_MIN_FLOP = float(_os.environ.get("PDFNET_ASYNC_WGRAD_MIN_GFLOP", "0")) * 1e9


def _param_grads(ctx, x, g, w, w_par, b_par, has_b, launch_w, C, R, flops, fused_bias=False, shadows=()):
    """Weight and bias gradients of a conv / transposed conv / linear layer.  Gradients that go straight into the
    trainer's flat buffer are issued together on the side stream (one stream switch for both).  A size threshold for the
    switch was measured (PDFNET_ASYNC_WGRAD_MIN_GFLOP = 0 / 0.5 / 4 -> 316 / 311 / 298 img/s): even the smallest layers
    gain from leaving the dependent main-stream chain, so the default is 0.
    `launch_w(out_w, out_b, accumulate)`; with `fused_bias` the weight-gradient launch also produces the bias gradient
    (out_b may be None), otherwise the bias gradient is a separate column-sum.  Returns (dw, db) for autograd (None when
    accumulated directly)."""
    need_w = ctx.needs_input_grad[1]
    need_b = has_b and ctx.needs_input_grad[2]
    mg_w = _main_grad(w_par, w) if need_w else None
    mg_b = _main_grad(b_par, b_par) if need_b else None
    dw = db = None
    ride = fused_bias and need_w and need_b and (mg_w is None) == (mg_b is None)      # same accumulate mode for both
    if mg_w is not None or mg_b is not None:
        # (the bf16 shadows the launch reads are side-stream inputs like x and g: recorded, or the allocator could hand their
        # memory out again while the weight-gradient kernel still reads it)
        def run():
            if mg_w is not None:
                launch_w(mg_w, mg_b if ride else None, 1)
            if mg_b is not None and not ride:
                _colsum_into(g, C, R, C, mg_b)
        if WGRAD_GROUP > 1 and ASYNC_WGRAD and USE_SIDE_STREAMS and flops >= ASYNC_WGRAD_MIN_FLOP:
            _wg_defer(run, (x, g) + tuple(shadows), (w_par, b_par))
        else:
            with wgrad_stream(flops >= ASYNC_WGRAD_MIN_FLOP, x, g, *shadows, params=(w_par, b_par)):
                run()
    if need_w and mg_w is None:
        dw = torch.empty_like(w)
        if ride:
            db = torch.empty(C, dtype=torch.float32, device=x.device)
        launch_w(dw, db, 0)
    if need_b and mg_b is None and not ride:
        db = _colsum(g, C, R, C)
    return dw, db


def _act_bwd(dy, y, act):
    R, C = _rows(y)
    g = torch.empty_like(y)
    _L().pdf_act_bwd(ptr(dy), C, ptr(y), C, ptr(g), C, C, R, act, stream())
    return g


# ----------------------------------------------------------------------------------------------
class _Conv2d(Function):
    """skip=True: also returns the input as a second output (the identity shortcut of a ResNet block).  The two gradients of
    the block input then meet in THIS backward, where the backward-data GEMM adds its result onto the shortcut's gradient in
    its epilogue (pdf_conv2d_bwd_data_add) -- instead of autograd's separate add pass over both tensors."""

    @staticmethod
    def forward(ctx, x, w, b, stride, pad, act, skip=False, stats=False):
        hip.require_gpu(x, w)
        w_in = w
        x_in = x
        x, w = cl(x), cl(w)
        N, Cin, H, W = x.shape
        Cout, _, KH, KW = w.shape
        OH = (H + 2 * pad - KH) // stride + 1
        OW = (W + 2 * pad - KW) // stride + 1
        y = torch.empty((N, Cout, OH, OW), dtype=torch.float32, device=x.device, memory_format=CL)
        x16, w16 = shadow_of(x), shadow_of(w)
        part = _stats_request(stats, N * OH * OW, Cout, x.device, batch=N)
        y16 = None
        if (stats and storage_on(N) and b is None and act == ACT_NONE and Cin % 16 == 0 and Cout % 16 == 0
                and (N * OH * OW) % 128 == 0):
            y16 = torch.empty_like(y, dtype=torch.bfloat16)  # the output: y itself stays unwritten (see BF16_STORAGE)
        #  F(4x4) carries ~4e-5 absolute error against 1e-6..3e-6 of the direct kernels: it is a TRAINING trade.  A forward that
        # nothing will be differentiated through (eval, no_grad) gets no workspace and therefore the direct kernels, unless asked otherwise.
        ws, nws = _wino_ws(N, H, W, Cin, Cout, KH, KW, stride, pad, 0, x.device) if (WINOGRAD_INFERENCE or any(ctx.needs_input_grad[:3])) else (None, None)
        # several convolutions reading ONE feature map (the heads on x0): the first F(4x4) forward leaves V, the transformed input, in its
        # workspace and the others take it from there (share_winograd_input) -- V depends on the input only
        share = getattr(x_in, '_pdf_wino_share', None) if (ws is not None and WINOGRAD_KEEP_V) else None
        v_off = _wino_v_offset(N, H, W, Cin, Cout, KH, KW, stride, pad) if (ws is not None and WINOGRAD_KEEP_V) else -1
        v_src = None
        if share is not None and v_off >= 0:
            ent = share.get((N, H, W, Cin))
            if ent is not None and ent[2] == x_in._version:
                v_src = ent
                if ent[4] != hip._raw_stream(hip._raw_device()):        # written on another stream: order after it, and tell the allocator
                    cs = torch.cuda.current_stream()
                    cs.wait_event(ent[3])
                    ent[0].record_stream(cs)
        o, oa = _O(op0_bf16=ptr(x16), op1_bf16=ptr(w16), out_bf16=ptr(y16), stats_out=ptr(part), stats_cap=part.numel() if part is not None else None,
                   ws=ptr(ws), ws_floats=nws, wino_v=(v_src[0].data_ptr() + 4 * v_src[1]) if v_src is not None else None)
        _L().pdf_conv2d_fwd_x(ptr(x), ptr(w), ptr(b), ptr(y), N, H, W, Cin, Cin, Cout, KH, KW, stride, pad, OH, OW, Cout, act, stream(), oa)
        if share is not None and v_off >= 0 and v_src is None:
            ev = torch.cuda.Event()
            ev.record()
            share[(N, H, W, Cin)] = (ws, v_off, x_in._version, ev, hip._raw_stream(hip._raw_device()))
        _stats_attach(y, part, o)
        if y16 is not None:
            if DEBUG_PHANTOMS:
                y.fill_(float('nan'))                       # any reader of the unwritten fp32 storage then shows up as NaN downstream
            y._pdf_y16 = (y16, y._version)
        # Winograd F(4x4): the weight gradient needs the same transformed input V the forward just wrote into its workspace -- keep the
        # workspace for it (one input-transform pass per layer and step less; V of `feat` at B=32 is 1.2 GB, all 17 layers ~5.5 GB)
        ctx.wino_v = None
        if v_off >= 0 and ctx.needs_input_grad[1]:
            ctx.wino_v = (v_src[0], v_src[1]) if v_src is not None else (ws, v_off)
        ctx.save_for_backward(x, w, y if act else None)
        ctx.s16 = (x16, w16)
        ctx.w16t = shadow_t_of(w) if w16 is not None else None
        ctx.cfg = (stride, pad, act, b is not None)
        ctx.params = (w_in, b)
        if skip:
            return y, x_in.view_as(x_in)
        return y

    @staticmethod
    def backward(ctx, dy, dskip=None):
        x, w, y = ctx.saved_tensors
        stride, pad, act, has_b = ctx.cfg
        N, Cin, H, W = x.shape
        Cout, _, KH, KW = w.shape
        OH, OW = dy.shape[2], dy.shape[3]
        g16 = _stored16(dy)                                  # bf16 storage mode: the gradient exists only as bf16 (dy is a phantom)
        if g16 is not None:
            g, gp = dy, None
        else:
            g = cl(dy)
            if act:
                g = _act_bwd(g, y, act)
            g16, gp = shadow_of(g), ptr(g)
        x16, w16 = ctx.s16
        dx = dw = db = None
        L = _L()
        if ctx.needs_input_grad[0]:
            ws, nws = _wino_ws(N, H, W, Cin, Cout, KH, KW, stride, pad, 1, x.device) if (OH == H and OW == W and gp is not None) else (None, None)
            _, oa = _O(op0_bf16=ptr(g16), op1_bf16=ptr(w16), op1_bf16_t=ptr(ctx.w16t) if g16 is not None else None, ws=ptr(ws), ws_floats=nws)
            if dskip is not None and stride == 1 and dskip.shape == x.shape and dskip.is_contiguous(memory_format=CL):
                dx = dskip                                  # the shortcut's gradient (sole consumer: this node); += in the epilogue
                L.pdf_conv2d_bwd_data_add_x(gp, ptr(w), ptr(dx), N, H, W, Cin, Cin, Cout, KH, KW, stride, pad, OH, OW, Cout, stream(), oa)
            else:
                dx = torch.zeros_like(x) if stride > KH else torch.empty_like(x)
                L.pdf_conv2d_bwd_data_x(gp, ptr(w), ptr(dx), N, H, W, Cin, Cin, Cout, KH, KW, stride, pad, OH, OW, Cout, stream(), oa)
                if dskip is not None:
                    dx = dx + dskip
        w_par, b_par = ctx.params
        R = N * OH * OW

        keep = ctx.wino_v
        ctx.wino_v = None

        def launch_w(out, out_b, acc):
            ws, n = _wgrad_ws(R, Cout, KH * KW * Cin, x.device)
            wws, nww = _wino_ws(N, H, W, Cin, Cout, KH, KW, stride, pad, 2, x.device) if (OH == H and OW == W and gp is not None) else (None, None)
            v = (keep[0].data_ptr() + 4 * keep[1]) if (keep is not None and wws is not None) else None
            L.pdf_conv2d_bwd_weight_x(ptr(x), gp, ptr(out), ptr(out_b), ptr(ws), n, N, H, W, Cin, Cin, Cout, KH, KW,
                                      stride, pad, OH, OW, Cout, acc, stream(), _O(op0_bf16=ptr(x16), op1_bf16=ptr(g16), ws=ptr(wws), ws_floats=nww, wino_v=v)[1])
        dw, db = _param_grads(ctx, x, g, w, w_par, b_par, has_b, launch_w, Cout, R, 2.0 * R * Cout * KH * KW * Cin, fused_bias=True,
                              shadows=(x16, g16, keep[0] if keep is not None else None))
        return dx, dw, db, None, None, None, None, None


def conv2d(x, w, b=None, stride=1, pad=0, act=ACT_NONE, stats=False):
    """stats=True: the output feeds a training-mode BatchNorm -- its statistics are taken in the GEMM epilogue (see
    _stats_request); harmless when the launch has no statistics epilogue (the BatchNorm then runs its own pass)."""
    return _Conv2d.apply(x, w, b, stride, pad, act, False, stats)


def conv2d_with_skip(x, w, b=None, stride=1, pad=0, act=ACT_NONE, stats=False):
    """-> (conv2d(x), x): use the second output as the block's identity shortcut (see _Conv2d)."""
    return _Conv2d.apply(x, w, b, stride, pad, act, True, stats)


class _Deconv2d(Function):
    """nn.ConvTranspose2d; w logical [Cin, Cout, KH, KW] in channels_last storage."""

    @staticmethod
    def forward(ctx, x, w, b, stride, pad):
        hip.require_gpu(x, w)
        w_in = w
        x, w = cl(x), cl(w)
        N, Cin, H, W = x.shape
        _, Cout, KH, KW = w.shape
        OH = (H - 1) * stride - 2 * pad + KH
        OW = (W - 1) * stride - 2 * pad + KW
        L = _L()
        y = torch.empty((N, Cout, OH, OW), dtype=torch.float32, device=x.device, memory_format=CL)
        x16, w16 = shadow_of(x), shadow_of(w)
        ws, nws = _x3_deconv_ws(N, H, W, Cin, Cout, KH, KW, stride, pad, 0, x.device)
        L.pdf_deconv2d_fwd_x(ptr(x), ptr(w), ptr(b), ptr(y), N, H, W, Cin, Cin, Cout, KH, KW, stride, pad, OH, OW, Cout, stream(),
                             _O(op0_bf16=ptr(x16), op1_bf16=ptr(w16), ws=ptr(ws), ws_floats=nws)[1])
        ctx.save_for_backward(x, w)
        ctx.s16 = (x16, w16)
        ctx.cfg = (stride, pad, b is not None)
        ctx.params = (w_in, b)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        stride, pad, has_b = ctx.cfg
        N, Cin, H, W = x.shape
        _, Cout, KH, KW = w.shape
        OH, OW = dy.shape[2], dy.shape[3]
        g = cl(dy)
        x16, w16 = ctx.s16
        g16 = shadow_of(g)
        L = _L()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            ws, nws = _x3_deconv_ws(N, H, W, Cin, Cout, KH, KW, stride, pad, 1, x.device)
            L.pdf_deconv2d_bwd_data_x(ptr(g), ptr(w), ptr(dx), N, H, W, Cin, Cin, Cout, KH, KW, stride, pad, OH, OW, Cout, stream(),
                                      _O(op0_bf16=ptr(g16), op1_bf16=ptr(w16), ws=ptr(ws), ws_floats=nws)[1])
        w_par, b_par = ctx.params

        def launch_w(out, out_b, acc):
            ws, n = _wgrad_ws(N * H * W, Cin, KH * KW * Cout, x.device)
            xws, nx = _x3_deconv_ws(N, H, W, Cin, Cout, KH, KW, stride, pad, 2, x.device)
            L.pdf_deconv2d_bwd_weight_x(ptr(x), ptr(g), ptr(out), ptr(ws), n, N, H, W, Cin, Cin, Cout, KH, KW,
                                        stride, pad, OH, OW, Cout, acc, stream(), _O(op0_bf16=ptr(x16), op1_bf16=ptr(g16), ws=ptr(xws), ws_floats=nx)[1])
        dw, db = _param_grads(ctx, x, g, w, w_par, b_par, has_b, launch_w, Cout, N * OH * OW, 2.0 * N * H * W * Cin * KH * KW * Cout, shadows=(x16, g16))
        return dx, dw, db, None, None


def deconv2d(x, w, b=None, stride=1, pad=0):
    return _Deconv2d.apply(x, w, b, stride, pad)


class _Linear(Function):
    """y[..., N] = act(x[..., K] w[N, K]^T + b)."""

    @staticmethod
    def forward(ctx, x, w, b, act, fp32=False, stats=False):
        hip.require_gpu(x, w)
        w_in = w
        lz = _lazy_bn(x)                                   # x = relu(BN(raw)) that was never written: the GEMM applies it while staging
        if lz is not None:
            x, aff = lz[0], (lz[1], lz[2])
        else:
            aff = None
        x, w = x.contiguous(), w.contiguous()
        K = x.shape[-1]
        M = x.numel() // K
        Nn = w.shape[0]
        y = torch.empty(x.shape[:-1] + (Nn,), dtype=torch.float32, device=x.device)
        x16, w16 = (None, None) if (fp32 or aff is not None) else (shadow_of(x), shadow_of(w))
        with _forced_fp32(fp32):
            part = _stats_request(stats, M, Nn, x.device)
            o, oa = _O(op0_bf16=ptr(x16), op1_bf16=ptr(w16), stats_out=ptr(part), stats_cap=part.numel() if part is not None else None,
                       in_scale=ptr(aff[0]) if aff is not None else None, in_shift=ptr(aff[1]) if aff is not None else None)
            _L().pdf_linear_fwd_x(ptr(x), ptr(w), ptr(b), ptr(y), M, Nn, K, K, K, Nn, act, stream(), oa)
            _stats_attach(y, part, o)
        ctx.save_for_backward(x, w, y if act else None, *(aff if aff is not None else ()))
        ctx.s16 = (x16, w16)
        ctx.w16t = shadow_t_of(w) if w16 is not None else None
        ctx.fp32 = fp32
        ctx.cfg = (act, b is not None)
        ctx.params = (w_in, b)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors[:3]
        aff = ctx.saved_tensors[3:5] if len(ctx.saved_tensors) > 3 else None
        act, has_b = ctx.cfg
        K = x.shape[-1]
        M = x.numel() // K
        Nn = w.shape[0]
        g = dy.contiguous()
        if act:
            g = _act_bwd(g, y, act)
        x16, w16 = ctx.s16
        g16 = None if ctx.fp32 else shadow_of(g)
        L = _L()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            with _forced_fp32(ctx.fp32):
                L.pdf_linear_bwd_data_x(ptr(g), ptr(w), ptr(dx), M, Nn, K, Nn, K, K, stream(),
                                        _O(op0_bf16=ptr(g16), op1_bf16=ptr(w16), op1_bf16_t=ptr(ctx.w16t) if g16 is not None else None)[1])
        w_par, b_par = ctx.params

        def launch_w(out, out_b, acc):
            ws, n = _wgrad_ws(M, Nn, K, x.device)
            with _forced_fp32(ctx.fp32):
                _, oa = _O(op0_bf16=ptr(x16), op1_bf16=ptr(g16), in_scale=ptr(aff[0]) if aff is not None else None,
                           in_shift=ptr(aff[1]) if aff is not None else None)
                L.pdf_linear_bwd_weight_x(ptr(x), ptr(g), ptr(out), ptr(out_b), ptr(ws), n, M, Nn, K, K, Nn, acc, stream(), oa)
        dw, db = _param_grads(ctx, x, g, w, w_par, b_par, has_b, launch_w, Nn, M, 2.0 * M * Nn * K, fused_bias=True, shadows=(x16, g16))
        return dx, dw, db, None, None, None


def linear(x, w, b=None, act=ACT_NONE, fp32=False, stats=False):
    """fp32=True: keep this layer on the exact fp32 kernels when the library runs in bf16 mode (layers whose INPUT must not
    be rounded -- absolute point coordinates that are only meaningful as differences).  stats: see conv2d."""
    return _Linear.apply(x, w, b, act, fp32, stats)


def as_matrix(weight):
    """A conv weight [Cout, Cin, KH, KW] in channels_last storage IS the row-major matrix [Cout, KH*KW*Cin]: return that
    view (no copy) for `linear`, carrying the same view of the trainer's flat gradient so that the weight gradient is still
    accumulated in place."""
    w2 = weight.permute(0, 2, 3, 1).reshape(weight.shape[0], -1)
    if w2.data_ptr() != weight.data_ptr():
        return w2                          # not channels_last storage: `reshape` copied; autograd carries dw back through it
    w16 = shadow_of(weight)
    if w16 is not None:
        s2 = w16.permute(0, 2, 3, 1).reshape(weight.shape[0], -1)
        if s2.data_ptr() == w16.data_ptr():
            attach_shadow(w2, s2)
    if getattr(weight, '_pdf_main_grad', False) and weight.grad is not None:
        # only when the re-shaped gradient is a VIEW of the flat gradient buffer: a copy would swallow the accumulation
        g2 = weight.grad.permute(0, 2, 3, 1).reshape(weight.shape[0], -1)
        if g2.data_ptr() == weight.grad.data_ptr():
            w2._pdf_main_grad = True
            w2._pdf_grad_alias = g2
    return w2


class _LinearPair(Function):
    """Two same-shaped Linear layers in one launch: x [2, ..., K]; half 0 goes through (w0, b0), half 1 through (w1, b1).
    The mesh decoder's left / right hand branches (DualGraph.py:83-84, inter_attn.py:66-67)."""

    @staticmethod
    def forward(ctx, x, w0, b0, w1, b1, act):
        hip.require_gpu(x, w0, w1)
        if x.shape[0] != 2 or w0.shape != w1.shape or (b0 is None) != (b1 is None):
            raise ValueError("pdfnet_amd: linear_pair wants x [2, ..., K] and two equal-shaped layers")
        w0_in, w1_in = w0, w1
        x, w0, w1 = x.contiguous(), w0.contiguous(), w1.contiguous()
        K = x.shape[-1]
        M = x.numel() // K // 2
        Nn = w0.shape[0]
        y = torch.empty(x.shape[:-1] + (Nn,), dtype=torch.float32, device=x.device)
        _L().pdf_linear_fwd_pair(ptr(x), ptr(w0), ptr(w1), ptr(b0), ptr(b1), ptr(y), M, Nn, K, K, K, Nn, act, stream())
        ctx.save_for_backward(x, w0, w1, y if act else None)
        ctx.cfg = (act, b0 is not None)
        ctx.params = (w0_in, b0, w1_in, b1)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w0, w1, y = ctx.saved_tensors
        act, has_b = ctx.cfg
        K = x.shape[-1]
        M = x.numel() // K // 2
        Nn = w0.shape[0]
        g = dy.contiguous()
        if act:
            g = _act_bwd(g, y, act)
        L = _L()
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            L.pdf_linear_bwd_data_pair(ptr(g), ptr(w0), ptr(w1), ptr(dx), M, Nn, K, Nn, K, K, stream())
        w0_par, b0_par, w1_par, b1_par = ctx.params
        need_w = ctx.needs_input_grad[1] or ctx.needs_input_grad[3]
        need_b = has_b and (ctx.needs_input_grad[2] or ctx.needs_input_grad[4])
        mg_w0, mg_w1 = (_main_grad(w0_par, w0), _main_grad(w1_par, w1)) if need_w else (None, None)
        mg_b0, mg_b1 = (_main_grad(b0_par, b0_par), _main_grad(b1_par, b1_par)) if need_b else (None, None)
        direct_w = mg_w0 is not None and mg_w1 is not None
        direct_b = mg_b0 is not None and mg_b1 is not None

        def launch_w(o0, o1, p0, p1, acc):
            n = 2 * L.pdf_wgrad_workspace_floats(M, Nn, K)
            L.pdf_linear_bwd_weight_pair(ptr(x), ptr(g), ptr(o0), ptr(o1), ptr(p0), ptr(p1), ptr(_ws(n, x.device)), n, M, Nn, K, K, Nn, acc, stream())

        def launch_b(o0, o1, acc):
            ws = _ws(2 * _bn_ws_floats(Nn, M), x.device)
            L.pdf_colsum_pair(ptr(g), Nn, Nn, M, ptr(o0), ptr(o1), acc, ptr(ws), stream())
        dw0 = dw1 = db0 = db1 = None
        ride = need_w and need_b and direct_w == direct_b                 # the bias gradients ride along with the weight launch
        if (need_w and direct_w) or (need_b and direct_b):
            with wgrad_stream(True, x, g, params=(w0_par, b0_par, w1_par, b1_par)):
                if need_w and direct_w:
                    launch_w(mg_w0, mg_w1, mg_b0 if ride else None, mg_b1 if ride else None, 1)
                if need_b and direct_b and not ride:
                    launch_b(mg_b0, mg_b1, 1)
        if need_w and not direct_w:
            dw0, dw1 = torch.empty_like(w0), torch.empty_like(w1)
            if ride:
                db0, db1 = torch.empty(Nn, device=x.device), torch.empty(Nn, device=x.device)
            launch_w(dw0, dw1, db0, db1, 0)
        if need_b and not direct_b and not ride:
            db0, db1 = torch.empty(Nn, device=x.device), torch.empty(Nn, device=x.device)
            launch_b(db0, db1, 0)
        return dx, dw0, db0, dw1, db1, None


def linear_pair(x, w0, b0, w1, b1, act=ACT_NONE):
    return _LinearPair.apply(x, w0, b0, w1, b1, act)


# ----------------------------------------------------------------------------------------------
class _BatchNorm(Function):
    """y = [relu]( BN(x) [+ res] ) over rows; x 4-D channels_last or [..., C]."""

    @staticmethod
    def forward(ctx, x, gamma, beta, rmean, rvar, res, training, momentum, eps, relu, lazy=False):
        hip.require_gpu(x)
        x16 = _stored16(x)                                  # bf16 storage mode: x is a phantom, its value is this bf16 tensor
        if x16 is not None and not training:
            raise RuntimeError("pdfnet_amd: a bf16-storage phantom reached a BatchNorm in eval mode -- the producing convolution must "
                               "be called with stats=<that BatchNorm>.training (only a training-mode BatchNorm reads the bf16 tensor)")
        if _lazy_bn(x) is not None:
            raise RuntimeError("pdfnet_amd: the unwritten output of a lazy BatchNorm reached another BatchNorm; only a linear layer may consume it")
        if x16 is None:
            x = _canon(x)
        res = _canon(res) if res is not None else None
        R, C = _rows(x)
        y = torch.empty_like(x)
        dev = x.device
        scale = torch.empty(C, device=dev)
        shift = torch.empty(C, device=dev)
        L = _L()
        if training:
            mean = torch.empty(C, device=dev)
            rstd = torch.empty(C, device=dev)
            tiles = tile_stats_of(x)
            if tiles is not None:                           # statistics came out of the producing GEMM's epilogue
                ws = None
            else:
                ws = _ws(_bn_ws_floats(C, R), dev)
            # lazy: statistics and (scale, shift) only -- the ONE consumer, a linear layer, applies BN + ReLU while it stages its rows
            # (PdfCallOpts::in_scale / in_shift) and y is never written: it only carries (x, scale, shift) to that consumer (_lazy_bn)
            lazy = bool(lazy) and relu and res is None and x16 is None and not _GEMM_BF16 and x.dim() == 2 and C % 16 == 0 and R % 16 == 0
            y16 = new_shadow(y) if (C % 4 == 0 and not lazy) else None
            _, oa = _O(out_bf16=ptr(y16), bn_x_bf16=ptr(x16), tile_stats=ptr(tiles[0]) if tiles is not None else None,
                       tile_n=tiles[1] if tiles is not None else None, tile_rows=tiles[2] if tiles is not None else None)
            L.pdf_bn_train_fwd_x(None if x16 is not None else ptr(x), C, C, R, ptr(gamma), ptr(beta), ptr(rmean), ptr(rvar), momentum, eps,
                                 ptr(res), C, int(relu), None if lazy else ptr(y), C, ptr(mean), ptr(rstd), ptr(scale), ptr(shift), ptr(ws), stream(), oa)
            if y16 is not None:
                attach_shadow(y, y16)
            if lazy:
                y._pdf_lazy = (x, scale, shift, y._version)
            # ReLU without a residual: the backward recomputes the mask from x with (scale, shift); y is not kept for it
            recompute = relu and res is None
            ctx.save_for_backward(x16 if x16 is not None else x, gamma, mean, rstd, (scale if recompute else (y if relu else None)), (shift if recompute else None))
            ctx.x_is_16 = x16 is not None
        else:
            L.pdf_bn_eval_fwd(ptr(x), C, C, R, ptr(gamma), ptr(beta), ptr(rmean), ptr(rvar), eps,
                              ptr(res), C, int(relu), ptr(y), C, ptr(scale), ptr(shift), stream())
            ctx.save_for_backward(x, gamma, None, None, y if relu else None, None)
            ctx.x_is_16 = False
        ctx.cfg = (training, relu, res is not None, eps)
        ctx.params = (gamma, beta)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, mean, rstd, y, shift = ctx.saved_tensors
        training, relu, has_res, eps = ctx.cfg
        scale = None
        mode = int(relu)
        if relu and not has_res:
            scale, y, mode = y, None, 2
        if not training:
            raise RuntimeError("pdfnet_amd: BatchNorm backward in eval mode is not implemented")
        R, C = _rows(x)
        g = _canon(dy)
        x_is_16 = ctx.x_is_16
        dx = torch.empty_like(x, dtype=torch.float32)       # (storage mode: allocated, not written -- the phantom autograd passes on)
        dres = torch.empty_like(dx) if has_res else None
        g_par, b_par = ctx.params
        mg_g, mg_b = _main_grad(g_par, g_par), _main_grad(b_par, b_par)
        direct = mg_g is not None and mg_b is not None
        dgamma = mg_g if direct else torch.empty(C, device=x.device)
        dbeta = mg_b if direct else torch.empty(C, device=x.device)
        L = _L()
        ws = _ws(_bn_ws_floats(C, R) + 3 * C, x.device)
        dx16 = new_shadow(dx) if C % 4 == 0 else None
        _, oa = _O(out_bf16=ptr(dx16), bn_x_bf16=ptr(x) if x_is_16 else None)
        L.pdf_bn_train_bwd_x(ptr(g), C, ptr(y), C, mode, None if x_is_16 else ptr(x), C, ptr(mean), ptr(rstd), ptr(gamma), ptr(scale), ptr(shift), C, R,
                             None if x_is_16 else ptr(dx), C, ptr(dres), C, ptr(dgamma), ptr(dbeta), int(direct), ptr(ws), stream(), oa)
        if x_is_16:
            if DEBUG_PHANTOMS:
                dx.fill_(float('nan'))
            dx._pdf_y16 = (dx16, dx._version)               # the conv's backward takes the bf16 gradient; dx itself was not written
        elif dx16 is not None:
            attach_shadow(dx, dx16)
        if direct:
            dgamma = dbeta = None
        return dx, dgamma, dbeta, None, None, dres, None, None, None, None, None


def batch_norm(x, gamma, beta, rmean, rvar, training, momentum=0.1, eps=1e-5, relu=False, res=None, lazy=False):
    """lazy=True (training, ReLU, no residual, fp32 mode, rows of C % 16 == 0 channels): the result may ONLY be fed to F.linear --
    it is never written; the linear layer applies the normalisation to its operand on the fly (forward and weight gradient)."""
    return _BatchNorm.apply(x, gamma, beta, rmean, rvar, res, training, momentum, eps, relu, lazy)


class _Pad2d(Function):
    """y [R2, C2] = x [R, C] in the top-left corner, zeros elsewhere (one launch; backward: the crop of dy, one launch)."""

    @staticmethod
    def forward(ctx, x, R2, C2):
        hip.require_gpu(x)
        x = x.contiguous()
        R, C = x.shape
        y = torch.empty((R2, C2), dtype=torch.float32, device=x.device)
        _L().pdf_pad2d(ptr(x), C, R, C, ptr(y), C2, R2, C2, stream())
        ctx.shape = (R, C)
        return y

    @staticmethod
    def backward(ctx, dy):
        R, C = ctx.shape
        dy = dy.contiguous()
        dx = torch.empty((R, C), dtype=torch.float32, device=dy.device)
        _L().pdf_pad2d(ptr(dy), dy.shape[1], dy.shape[0], dy.shape[1], ptr(dx), C, R, C, stream())
        return dx, None, None


def pad2d(x, rows, cols):
    """Zero-pad a 2-D fp32 tensor to [rows, cols] (no-op when it already has that shape)."""
    if x.shape[0] == rows and x.shape[1] == cols:
        return x
    return _Pad2d.apply(x, rows, cols)


class _Act(Function):
    @staticmethod
    def forward(ctx, x, act):
        hip.require_gpu(x)
        x = _canon(x)
        R, C = _rows(x)
        y = torch.empty_like(x)
        _L().pdf_act_fwd(ptr(x), C, ptr(y), C, C, R, act, stream())
        ctx.save_for_backward(y)
        ctx.act = act
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return _act_bwd(_canon(dy), y, ctx.act), None


def relu(x):
    return _Act.apply(x, ACT_RELU)


class _MaxPool3s2(Function):
    """skip=True: also returns the input as a second output for its other consumers (see _Conv2d): their gradient arrives here
    and the pooling gradient is added onto it in the same pass (pdf_maxpool3s2_bwd_add)."""

    @staticmethod
    def forward(ctx, x, skip=False):
        x_in = x
        x = cl(x)
        N, C, H, W = x.shape
        OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        y = torch.empty((N, C, OH, OW), device=x.device, memory_format=CL)
        arg = torch.empty(y.numel(), dtype=torch.uint8, device=x.device)
        _L().pdf_maxpool3s2_fwd(ptr(x), N, H, W, C, ptr(y), ptr(arg), stream())
        ctx.save_for_backward(arg)
        ctx.shape = (N, C, H, W)
        ctx.set_materialize_grads(False)
        if skip:
            return y, x_in.view_as(x_in)
        return y

    @staticmethod
    def backward(ctx, dy, dskip=None):
        (arg,) = ctx.saved_tensors
        N, C, H, W = ctx.shape
        if dy is None:
            return dskip, None
        if dskip is not None and dskip.shape == (N, C, H, W) and dskip.is_contiguous(memory_format=CL) and dskip.dtype == torch.float32:
            _L().pdf_maxpool3s2_bwd_add(ptr(cl(dy)), ptr(arg), N, H, W, C, ptr(dskip), stream())
            return dskip, None
        # (C % 4 == 0: the gather kernel writes every element once; otherwise the atomic scatter needs a zero-filled dx)
        dx = torch.empty((N, C, H, W), device=dy.device, memory_format=CL) if C % 4 == 0 else _zeros_cl((N, C, H, W), dy.device)
        _L().pdf_maxpool3s2_bwd(ptr(cl(dy)), ptr(arg), N, H, W, C, ptr(dx), stream())
        if dskip is not None:
            dx = dx + dskip
        return dx, None


def maxpool3s2(x, skip=False):
    return _MaxPool3s2.apply(x, skip)


class _Up2(Function):
    @staticmethod
    def forward(ctx, x):
        x = cl(x)
        N, C, H, W = x.shape
        y = torch.empty((N, C, 2 * H, 2 * W), device=x.device, memory_format=CL)
        _L().pdf_upsample2x_fwd(ptr(x), N, H, W, C, ptr(y), stream())
        ctx.shape = (N, C, H, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        N, C, H, W = ctx.shape
        dx = torch.empty((N, C, H, W), dtype=torch.float32, device=dy.device, memory_format=CL)
        _L().pdf_upsample2x_bwd(ptr(cl(dy)), N, H, W, C, ptr(dx), stream())
        return dx


def upsample2x(x):
    return _Up2.apply(x)


class _L2Norm(Function):
    @staticmethod
    def forward(ctx, x, w, eps):
        x = cl(x)
        R, C = _rows(x)
        y = torch.empty_like(x)
        norm = torch.empty(R, device=x.device)
        _L().pdf_l2norm_fwd(ptr(x), C, C, R, ptr(w), eps, ptr(y), C, ptr(norm), stream())
        ctx.save_for_backward(x, w, norm)
        ctx.eps = eps
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, norm = ctx.saved_tensors
        R, C = _rows(x)
        dx = torch.empty_like(x)
        dw = torch.zeros_like(w)
        _L().pdf_l2norm_bwd(ptr(cl(dy)), C, ptr(x), C, C, R, ptr(w), ctx.eps, ptr(norm), ptr(dx), C, ptr(dw), stream())
        return dx, dw, None


def l2norm(x, w, eps=1e-10):
    return _L2Norm.apply(x, w, eps)


def _ptr_array(tensors):
    import ctypes
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def _int_array(values):
    import ctypes
    return (ctypes.c_int * len(values))(*values)


class _L2NormCat(Function):
    """torch.cat([l2norm(x_i, w_i)], 1) without the concatenation pass: every L2Norm writes its channels straight into the
    concatenated NHWC buffer (row stride = the total channel count), and the backward reads its channel slice of the incoming
    gradient in place (no slice copies).  The pyramid of intaghand_encoder.py:724-739."""

    @staticmethod
    def forward(ctx, eps, *args):
        n = len(args) // 2
        xs, ws = [cl(x) for x in args[:n]], list(args[n:])
        hip.require_gpu(*xs)
        B, _, H, W = xs[0].shape
        if any(x.shape[0] != B or x.shape[2:] != xs[0].shape[2:] for x in xs):
            raise ValueError("pdfnet_amd: l2norm_cat wants maps of one batch and resolution")
        Cs = [x.shape[1] for x in xs]
        Ct, R = sum(Cs), B * H * W
        out = torch.empty((B, Ct, H, W), dtype=torch.float32, device=xs[0].device, memory_format=CL)
        norms = [torch.empty(R, device=out.device) for _ in xs]
        out16 = new_shadow(out) if all(C % 64 == 0 for C in Cs) else None
        if out16 is not None:
            attach_shadow(out, out16)
        _L().pdf_l2norm_cat_fwd_x(n, _ptr_array(xs), _int_array(Cs), _ptr_array(ws), eps, R, ptr(out), Ct, _ptr_array(norms), stream(), _O(out_bf16=ptr(out16))[1])
        ctx.save_for_backward(*xs, *ws, *norms)
        ctx.cfg = (eps, n, Cs)
        return out

    @staticmethod
    def backward(ctx, dy):
        eps, n, Cs = ctx.cfg
        t = ctx.saved_tensors
        xs, ws, norms = t[:n], t[n:2 * n], t[2 * n:]
        g = cl(dy)
        Ct = sum(Cs)
        R = xs[0].numel() // Cs[0]
        dxs, dws = [torch.empty_like(x) for x in xs], [torch.zeros_like(w) for w in ws]
        d16 = [new_shadow(d) for d in dxs] if all(C % 64 == 0 for C in Cs) and _os.environ.get('PDFNET_L2_SHADOW', '1') != '0' else None
        if d16 is not None and d16[0] is None:
            d16 = None
        _L().pdf_l2norm_cat_bwd(n, ptr(g), Ct, _ptr_array(xs), _int_array(Cs), _ptr_array(ws), eps, R, _ptr_array(norms),
                                _ptr_array(dxs), _ptr_array(dws), _ptr_array(d16) if d16 is not None else None, stream())
        if d16 is not None:
            for d, s16 in zip(dxs, d16):
                attach_shadow(d, s16)
        return (None,) + tuple(dxs) + tuple(dws)


def l2norm_cat(xs, ws, eps=1e-10):
    return _L2NormCat.apply(eps, *xs, *ws)


class _LayerNorm(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        hip.require_gpu(x)
        x = x.contiguous()
        R, Fd = x.numel() // x.shape[-1], x.shape[-1]
        y = torch.empty_like(x)
        mean = torch.empty(R, device=x.device)
        rstd = torch.empty(R, device=x.device)
        _L().pdf_layernorm_fwd(ptr(x), Fd, Fd, R, ptr(gamma), ptr(beta), eps, ptr(y), Fd, ptr(mean), ptr(rstd), stream())
        ctx.save_for_backward(x, gamma, mean, rstd)
        ctx.params = (gamma, beta)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, mean, rstd = ctx.saved_tensors
        R, Fd = x.numel() // x.shape[-1], x.shape[-1]
        dx = torch.empty_like(x)
        g_par, b_par = ctx.params
        mg_g, mg_b = _main_grad(g_par, g_par), _main_grad(b_par, b_par)
        direct = mg_g is not None and mg_b is not None
        dg = mg_g if direct else torch.zeros_like(gamma)
        db = mg_b if direct else torch.zeros_like(gamma)
        _L().pdf_layernorm_bwd(ptr(dy.contiguous()), Fd, ptr(x), Fd, Fd, R, ptr(gamma), ptr(mean), ptr(rstd), ptr(dx), Fd, ptr(dg), ptr(db), stream())
        if direct:
            dg = db = None
        return dx, dg, db, None


def layer_norm(x, gamma, beta, eps=1e-6):
    return _LayerNorm.apply(x, gamma, beta, eps)


class _LayerNormFused(Function):
    """z = x + dropout_p(add) (optional);  y = act(LayerNorm(z) * gamma_g + beta_g), g = first / second half of the rows when a
    second parameter set is given (x stacked [2, ...]).  Returns y, or (z, y) when `add` is given (z is the residual stream
    the caller keeps using)."""

    @staticmethod
    def forward(ctx, x, add, g0, b0, g1, b1, eps, act, p, seed):
        hip.require_gpu(x)
        x = x.contiguous()
        R, Fd = x.numel() // x.shape[-1], x.shape[-1]
        pair = g1 is not None
        if pair and (x.shape[0] != 2):
            raise ValueError("pdfnet_amd: paired layer_norm wants x stacked [2, ...]")
        split = R // 2 if pair else R
        y = torch.empty_like(x)
        mean = torch.empty(R, device=x.device)
        rstd = torch.empty(R, device=x.device)
        z = None
        if add is not None:
            add = add.contiguous()
            z = torch.empty_like(x)
        _L().pdf_layernorm_fused_fwd(ptr(x), Fd, ptr(add), Fd, p, seed, ptr(step_counter(x.device)), Fd, R, split,
                                     ptr(g0), ptr(b0), ptr(g1 if pair else g0), ptr(b1 if pair else b0), eps, act,
                                     ptr(z), Fd, ptr(y), Fd, ptr(mean), ptr(rstd), stream())
        ctx.save_for_backward(x if z is None else z, g0, g1, mean, rstd, y if act else None)
        ctx.cfg = (act, p, seed, add is not None, pair)
        ctx.params = (g0, b0, g1, b1)
        ctx.set_materialize_grads(False)
        if z is None:
            return y
        return z, y

    @staticmethod
    def backward(ctx, *grads):
        zin, g0, g1, mean, rstd, y = ctx.saved_tensors
        act, p, seed, has_add, pair = ctx.cfg
        dz_in, dy = grads if has_add else (None, grads[0])
        R, Fd = zin.numel() // zin.shape[-1], zin.shape[-1]
        split = R // 2 if pair else R
        if dy is None:                                       # only the residual stream was used downstream
            if dz_in is None:
                return (None,) * 10
            dadd = _Dropout.apply(dz_in, p, seed, True) if p > 0 else dz_in
            return dz_in, dadd, None, None, None, None, None, None, None, None
        pars = ctx.params
        mgs = [_main_grad(t, t) if t is not None else None for t in pars]
        need = [t is not None for t in pars]
        direct = all(m is not None for m, n in zip(mgs, need) if n)
        outs = [m if direct else (torch.zeros_like(t) if t is not None else None) for m, t in zip(mgs, pars)]
        dz = torch.empty_like(zin)
        dadd = torch.empty_like(zin) if has_add else None
        d0, e0 = outs[0], outs[1]
        d1, e1 = (outs[2], outs[3]) if pair else (d0, e0)
        dy = dy.contiguous()
        dz_in = dz_in.contiguous() if dz_in is not None else None
        gb = ptr(g1 if pair else g0)
        if direct and ASYNC_WGRAD and USE_SIDE_STREAMS:
            # data gradient on the dependent chain (no atomics), parameter gradients into the flat buffer on the weight-gradient
            # side stream: the one-shot kernel's same-address atomics were most of its 28 us, 37 times on the decoder's chain
            _L().pdf_layernorm_fused_bwd(ptr(dy), Fd, ptr(y), Fd, act, ptr(zin), Fd, Fd, R, split, ptr(g0), gb, ptr(mean), ptr(rstd),
                                         ptr(dz_in), Fd, ptr(dz), Fd, ptr(dadd), Fd, p, seed, ptr(step_counter(zin.device)),
                                         None, None, None, None, stream())
            with wgrad_stream(True, dy, y, zin, mean, rstd, params=pars):
                _L().pdf_layernorm_fused_bwd(ptr(dy), Fd, ptr(y), Fd, act, ptr(zin), Fd, Fd, R, split, ptr(g0), gb, ptr(mean), ptr(rstd),
                                             None, Fd, None, Fd, None, Fd, 0.0, 0, None, ptr(d0), ptr(e0), ptr(d1), ptr(e1), stream())
        else:
            _L().pdf_layernorm_fused_bwd(ptr(dy), Fd, ptr(y), Fd, act, ptr(zin), Fd, Fd, R, split, ptr(g0), gb,
                                         ptr(mean), ptr(rstd), ptr(dz_in), Fd, ptr(dz), Fd,
                                         ptr(dadd), Fd, p, seed, ptr(step_counter(zin.device)), ptr(d0), ptr(e0), ptr(d1), ptr(e1), stream())
        if direct:
            outs = [None] * 4
        return dz, dadd, outs[0], outs[1], outs[2], outs[3], None, None, None, None


def layer_norm_fused(x, gamma, beta, eps=1e-6, act=ACT_NONE, add=None, p=0.0, training=False, gamma1=None, beta1=None):
    """See _LayerNormFused.  `add` is the operand that goes through dropout(p) before the sum."""
    p = float(p) if (training and add is not None) else 0.0
    return _LayerNormFused.apply(x, add, gamma, beta, gamma1, beta1, eps, act, p, next_seed() if p > 0 else 0)


class _Dropout(Function):
    @staticmethod
    def forward(ctx, x, p, seed, _unused=None):
        x = x.contiguous()
        y = torch.empty_like(x)
        _L().pdf_dropout(ptr(x), ptr(y), x.numel(), p, seed, ptr(step_counter(x.device)), stream())
        ctx.cfg = (p, seed)
        return y

    @staticmethod
    def backward(ctx, dy):
        p, seed = ctx.cfg
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        _L().pdf_dropout(ptr(dy), ptr(dx), dy.numel(), p, seed, ptr(step_counter(dy.device)), stream())
        return dx, None, None, None


class _DropoutAdd(Function):
    """res + dropout(x)."""

    @staticmethod
    def forward(ctx, x, res, p, seed):
        x, res = x.contiguous(), res.contiguous()
        y = torch.empty_like(x)
        _L().pdf_dropout_add(ptr(x), ptr(res), ptr(y), x.numel(), p, seed, ptr(step_counter(x.device)), stream())
        ctx.cfg = (p, seed)
        return y

    @staticmethod
    def backward(ctx, dy):
        p, seed = ctx.cfg
        dx = dy
        if p > 0 and ctx.needs_input_grad[0]:
            dy = dy.contiguous()
            dx = torch.empty_like(dy)
            _L().pdf_dropout(ptr(dy), ptr(dx), dy.numel(), p, seed, ptr(step_counter(dy.device)), stream())
        return dx, dy, None, None


def dropout_add(x, res, p, training):
    """res + dropout(x, p)."""
    p = float(p) if training else 0.0
    return _DropoutAdd.apply(x, res, p, next_seed() if p > 0 else 0)


_seed_state = [0x5DEECE66D]


def next_seed():
    """Host-side counter; each dropout site of each step gets a distinct mask stream."""
    _seed_state[0] = (_seed_state[0] * 6364136223846793005 + 1442695040888963407) & ((1 << 63) - 1)
    return _seed_state[0]


_step_counters = {}


def step_counter(dev):
    """Device-resident step counter mixed into every dropout seed.  The trainer increments it inside the
    captured step, so hipGraph replays draw fresh masks although the host-side seeds are baked in."""
    key = str(dev)
    if key not in _step_counters:
        _step_counters[key] = torch.zeros(1, dtype=torch.int64, device=dev)
    return _step_counters[key]


def manual_seed(s):
    _seed_state[0] = (int(s) * 2654435761 + 12345) & ((1 << 63) - 1)


def dropout(x, p, training):
    if not training or p <= 0.0:
        return x
    return _Dropout.apply(x, float(p), next_seed())


# ----------------------------------------------------------------------------------------------
class _SFTModulate(Function):
    """fea * (scale + 1) + shift   on [..., C] rows."""

    @staticmethod
    def forward(ctx, fea, scale, shift):
        fea, scale, shift = fea.contiguous(), scale.contiguous(), shift.contiguous()
        R, C = _rows(fea)
        out = torch.empty_like(fea)
        _L().pdf_sft_fwd(ptr(fea), C, ptr(scale), C, ptr(shift), C, ptr(out), C, C, R, stream())
        ctx.save_for_backward(fea, scale)
        return out

    @staticmethod
    def backward(ctx, g):
        fea, scale = ctx.saved_tensors
        g = g.contiguous()
        R, C = _rows(fea)
        dfea = torch.empty_like(fea)
        dscale = torch.empty_like(fea)
        _L().pdf_sft_bwd(ptr(g), C, ptr(fea), C, ptr(scale), C, ptr(dfea), C, ptr(dscale), C, C, R, stream())
        return dfea, dscale, g


def sft_modulate(fea, scale, shift):
    return _SFTModulate.apply(fea, scale, shift)


class _SFT3(Function):
    """SFTLayer(3, 3) in one launch per direction (pdf_sft3_fwd / pdf_sft3_bwd): fea, cond [..., 3]; params = the layer's
    (weight [3,3,1,1], bias [3]) pairs in the order scale_conv0, scale_conv1, shift_conv0, shift_conv1."""

    @staticmethod
    def forward(ctx, fea, cond, *params):
        hip.require_gpu(fea, cond)
        fea, cond = fea.contiguous(), cond.contiguous()
        if fea.shape[-1] != 3 or cond.shape[-1] != 3 or len(params) != 8:
            raise ValueError("pdfnet_amd: sft3 wants 3-channel rows and four (weight, bias) pairs")
        pc = [p.contiguous() for p in params]
        R = fea.numel() // 3
        out = torch.empty_like(fea)
        _L().pdf_sft3_fwd(ptr(fea), 3, ptr(cond), 3, *[ptr(p) for p in pc], ptr(out), 3, R, stream())
        ctx.save_for_backward(fea, cond, *pc)
        ctx.params = params
        return out

    @staticmethod
    def backward(ctx, g):
        fea, cond = ctx.saved_tensors[:2]
        pc = ctx.saved_tensors[2:]
        g = g.contiguous()
        R = fea.numel() // 3
        dfea = torch.empty_like(fea)
        dcond = torch.empty_like(cond) if ctx.needs_input_grad[1] else None
        ws = _ws(48 * 256, g.device)
        mg = [_main_grad(p, p) if ctx.needs_input_grad[2 + i] else None for i, p in enumerate(ctx.params)]
        direct = all(m is not None for m in mg)
        if direct:                                             # straight into the trainer's flat gradient buffer
            outs, ret = mg, [None] * 8
        else:
            outs = [torch.empty_like(p) if ctx.needs_input_grad[2 + i] else None for i, p in enumerate(pc)]
            ret = outs
        _L().pdf_sft3_bwd(ptr(g), 3, ptr(fea), 3, ptr(cond), 3, *[ptr(p) for p in pc], ptr(dfea), 3, ptr(dcond), 3,
                          *[ptr(o) for o in outs], 1 if direct else 0, ptr(ws), R, stream())
        return (dfea, dcond) + tuple(ret)


def sft3(fea, cond, params):
    return _SFT3.apply(fea, cond, *params)


class _GatherRows(Function):
    """feat 4-D channels_last [B,C,H,W]; ind int64 [B,M] -> [B,M,ldo] (channels >= C zero)."""

    @staticmethod
    def forward(ctx, feat, ind, R, shift, ldo, chain=False):
        hip.require_gpu(feat, ind)
        feat_in = feat
        feat = cl(feat)
        B, C, H, W = feat.shape
        M = ind.shape[1]
        assert ind.dtype == torch.int64 and ind.stride(1) == 1
        out = torch.empty((B, M, ldo), device=feat.device)
        _L().pdf_gather_rows(ptr(feat), C, C, H * W, ptr(ind), ind.stride(0), B, M, R, shift, ptr(out), ldo, stream())
        ctx.save_for_backward(ind)
        ctx.cfg = (B, C, H, W, M, R, shift, ldo)
        ctx.set_materialize_grads(False)                     # an unused alias arrives as None, not as an NCHW-strided zero tensor
        if chain:
            return out, feat_in.view_as(feat_in)
        return out

    @staticmethod
    def backward(ctx, g, dalias=None):
        (ind,) = ctx.saved_tensors
        B, C, H, W, M, R, shift, ldo = ctx.cfg
        if g is None:
            return dalias, None, None, None, None, None
        dfeat, extra = _chain_target(dalias, (B, C, H, W), g.device)
        _L().pdf_scatter_rows_add(ptr(g.contiguous()), ldo, C, H * W, ptr(ind), ind.stride(0), B, M, R, shift, ptr(dfeat), C, stream())
        if extra is not None:
            dfeat = dfeat + extra
        return dfeat, None, None, None, None, None


def _chain_target(dalias, shape, dev):
    """Where a scatter-type backward accumulates: `dalias` itself -- the gradient of the feature map's later consumers, handed
    over through the alias output of a `chain=True` gather (a zero tensor from autograd when there are none) -- so that the
    consumers of one map fill ONE gradient tensor instead of a zero-filled tensor each plus an add pass per pair.
    -> (target, tensor still to be added or None)."""
    if dalias is None:
        return _zeros_cl(shape, dev), None
    if tuple(dalias.shape) == tuple(shape) and dalias.dtype == torch.float32 and dalias.is_contiguous(memory_format=CL):
        return dalias, None
    return _zeros_cl(shape, dev), dalias


def gather_rows(feat, ind, R=1, shift=0, ldo=None, chain=False):
    """chain=True -> (rows, feat): pass the second output to the map's NEXT consumer (see _chain_target)."""
    return _GatherRows.apply(feat, ind, R, shift, ldo or feat.shape[1], chain)


class _WindowGather(Function):
    """feat [B,C,H,W] channels_last, ind [B,M] -> windows [B*M, C, win, win] (channels_last), zero outside."""

    @staticmethod
    def forward(ctx, feat, ind, r, chain=False):
        hip.require_gpu(feat, ind)
        feat_in = feat
        feat = cl(feat)
        B, C, H, W = feat.shape
        M, win = ind.shape[1], 2 * r + 1
        out = torch.empty((B * M, C, win, win), device=feat.device, memory_format=CL)
        _L().pdf_window_op(ptr(feat), C, C, H, W, ptr(ind), ind.stride(0), B, M, r, ptr(out), None, 0, stream())
        ctx.save_for_backward(ind)
        ctx.cfg = (B, C, H, W, M, r)
        ctx.set_materialize_grads(False)
        if chain:
            return out, feat_in.view_as(feat_in)
        return out

    @staticmethod
    def backward(ctx, g, dalias=None):
        (ind,) = ctx.saved_tensors
        B, C, H, W, M, r = ctx.cfg
        if g is None:
            return dalias, None, None, None
        dfeat, extra = _chain_target(dalias, (B, C, H, W), g.device)
        _L().pdf_window_op(ptr(dfeat), C, C, H, W, ptr(ind), ind.stride(0), B, M, r, ptr(cl(g)), None, 1, stream())
        if extra is not None:
            dfeat = dfeat + extra
        return dfeat, None, None, None


class _WindowMask(Function):
    """x [B*M, C, win, win]: zero the window positions that fall outside the H x W map (its own backward)."""

    @staticmethod
    def forward(ctx, x, ind, H, W, r):
        x = cl(x)
        C = x.shape[1]
        B, M = ind.shape
        y = torch.empty_like(x)
        _L().pdf_window_op(None, C, C, H, W, ptr(ind), ind.stride(0), B, M, r, ptr(y), ptr(x), 2, stream())
        ctx.save_for_backward(ind)
        ctx.cfg = (H, W, r)
        return y

    @staticmethod
    def backward(ctx, g):
        (ind,) = ctx.saved_tensors
        H, W, r = ctx.cfg
        return _WindowMask.apply(g, ind, H, W, r), None, None, None, None


def window_gather(feat, ind, r, chain=False):
    return _WindowGather.apply(feat, ind, r, chain)


def window_mask(x, ind, H, W, r):
    return _WindowMask.apply(x, ind, H, W, r)


class _KnnGroup(Function):
    """pts [Bc,N,ldp] rows (C real channels, xyz first) -> grouped [Bc,S,K,ldg], idx [Bc,S,K] int32."""

    @staticmethod
    def forward(ctx, pts, C, S, K, r2, ldg):
        hip.require_gpu(pts)
        pts = pts.contiguous()
        Bc, N, ldp = pts.shape
        idx = torch.empty((Bc, S, K), dtype=torch.int32, device=pts.device)
        g = torch.empty((Bc, S, K, ldg), device=pts.device)
        _L().pdf_knn_ball_group(ptr(pts), ldp, C, Bc, N, S, K, r2, ptr(idx), ptr(g), ldg, stream())
        ctx.save_for_backward(idx)
        ctx.cfg = (Bc, N, ldp, C, S, K, ldg)
        ctx.mark_non_differentiable(idx)
        return g, idx

    @staticmethod
    def backward(ctx, dg, _):
        (idx,) = ctx.saved_tensors
        Bc, N, ldp, C, S, K, ldg = ctx.cfg
        dpts = torch.zeros((Bc, N, ldp), device=dg.device)
        _L().pdf_group_bwd(ptr(dg.contiguous()), ldg, ptr(idx), ptr(dpts), ldp, C, Bc, N, S, K, stream())
        return dpts, None, None, None, None, None


def knn_ball_group(pts, C, S, K, r2, ldg):
    return _KnnGroup.apply(pts, C, S, K, float(r2), ldg)


def knn_ball_indices(pts, S, K, r2):
    """Neighbour indices only (kNN + ball rule of group_points, lib/utils/utils.py:140-151): pts [Bc,N,>=3] -> int32 [Bc,S,K]."""
    hip.require_gpu(pts)
    p = pts.detach().contiguous()
    Bc, N, ldp = p.shape
    idx = torch.empty((Bc, S, K), dtype=torch.int32, device=p.device)
    _L().pdf_knn_ball_group(ptr(p), ldp, 3, Bc, N, S, K, float(r2), ptr(idx), None, 0, stream())
    return idx


class _GatherSub(Function):
    """y[b,s,k,:] = u[b, idx[b,s,k], :] - v[b,s,:]   (u [Bc,N,C], v [Bc,S,C], idx int32 [Bc,S,K]) -> [Bc,S,K,C]."""

    @staticmethod
    def forward(ctx, u, v, idx):
        hip.require_gpu(u, v, idx)
        u, v = u.contiguous(), v.contiguous()
        Bc, N, C = u.shape
        S, K = idx.shape[1], idx.shape[2]
        y = torch.empty((Bc, S, K, C), device=u.device)
        L = _L()
        L.pdf_gather_sub_fwd(ptr(u), C, ptr(v), C, ptr(idx), Bc, N, S, K, C, ptr(y), C, stream())
        inv = None
        if GATHER_SORTED and (u.requires_grad or v.requires_grad) and (2 * N + 1) * 4 <= 160 * 1024:
            # the index inverted now, beside the forward (it is launch-bound filler there), for a backward without float atomics
            start = torch.empty((Bc, N + 1), dtype=torch.int32, device=u.device)
            lst = torch.empty((Bc, S * K), dtype=torch.int32, device=u.device)
            L.pdf_invert_index(ptr(idx), Bc, N, S * K, ptr(start), ptr(lst), None, stream())
            inv = (start, lst)
        ctx.save_for_backward(idx, *(inv or ()))
        ctx.cfg = (Bc, N, S, K, C)
        return y

    @staticmethod
    def backward(ctx, dy):
        idx = ctx.saved_tensors[0]
        Bc, N, S, K, C = ctx.cfg
        dv = torch.empty((Bc, S, C), device=dy.device)
        if len(ctx.saved_tensors) == 3:                      # deterministic: every point sums its rows of dy in a fixed order
            start, lst = ctx.saved_tensors[1:]
            du = torch.empty((Bc, N, C), device=dy.device)
            _L().pdf_gather_sub_bwd_sorted(ptr(dy.contiguous()), C, ptr(start), ptr(lst), ptr(du), C, ptr(dv), C, Bc, N, S, K, C, stream())
        else:
            du = torch.zeros((Bc, N, C), device=dy.device)
            _L().pdf_gather_sub_bwd(ptr(dy.contiguous()), C, ptr(idx), ptr(du), C, ptr(dv), C, Bc, N, S, K, C, stream())
        return du, dv, None


GATHER_SORTED = _os.environ.get("PDFNET_GATHER_SORTED", "1") != "0"


def gather_sub(u, v, idx):
    return _GatherSub.apply(u, v, idx)


def fps(xyz, S, start=None):
    """Farthest point sampling (the reference's `farthest_point_sampling_fast`, lib/datasets/interhand.py:147-178):
    xyz [Bc,N,>=3] -> int32 [Bc,S] picks in order; start int32 [Bc] (first pick, default 0).  No gradient."""
    hip.require_gpu(xyz)
    x = xyz.detach().contiguous()
    Bc, N, ld = x.shape
    idx = torch.empty((Bc, S), dtype=torch.int32, device=x.device)
    st = start.to(torch.int32).contiguous() if start is not None else None
    _L().pdf_fps(ptr(x), ld, Bc, N, S, ptr(st), ptr(idx), stream())
    return idx


class _MaxK(Function):
    """x [R, K, C] -> max over K -> [R, C]."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        R, K, C = x.shape
        y = torch.empty((R, C), device=x.device)
        arg = torch.empty((R, C), dtype=torch.int32, device=x.device)
        _L().pdf_maxk_fwd(ptr(x), C, C, R, K, ptr(y), C, ptr(arg), stream())
        ctx.save_for_backward(arg)
        ctx.cfg = (R, K, C)
        return y

    @staticmethod
    def backward(ctx, dy):
        (arg,) = ctx.saved_tensors
        R, K, C = ctx.cfg
        dx = torch.empty((R, K, C), device=dy.device)
        _L().pdf_maxk_bwd(ptr(dy.contiguous()), C, ptr(arg), C, R, K, ptr(dx), C, stream())
        return dx


def max_over_k(x):
    return _MaxK.apply(x)


class _BnReluMaxK(Function):
    """BatchNorm -> ReLU -> max over the K neighbours of x [R*K, C] (rows r*K + k) in one pass: -> [R, C].
    The normalised [R*K, C] tensor is never materialised (set-abstraction tail, intaghand_encoder.py:59-62,79-82,97-100)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, rmean, rvar, K, training, momentum, eps):
        hip.require_gpu(x)
        x = x.contiguous()
        C = x.shape[-1]
        R = x.numel() // (C * K)
        dev = x.device
        out = torch.empty((R, C), device=dev)
        arg = torch.empty((R, C), dtype=torch.int32, device=dev)
        mean, rstd, scale, shift = (torch.empty(C, device=dev) for _ in range(4))
        L = _L()
        tiles = tile_stats_of(x) if training else None
        if tiles is not None:
            ws = None
        else:
            ws = _ws(_bn_ws_floats(C, R * K), dev)
        _, oa = _O(tile_stats=ptr(tiles[0]) if tiles is not None else None, tile_n=tiles[1] if tiles is not None else None,
                   tile_rows=tiles[2] if tiles is not None else None)
        L.pdf_bn_relu_maxk_fwd_x(ptr(x), C, C, R, K, ptr(gamma), ptr(beta), ptr(rmean), ptr(rvar), momentum, eps, int(training),
                                 ptr(out), C, ptr(arg), ptr(mean), ptr(rstd), ptr(scale), ptr(shift), ptr(ws), stream(), oa)
        ctx.save_for_backward(x, gamma, arg, mean, rstd, scale, shift)
        ctx.cfg = (R, K, C, training)
        ctx.params = (gamma, beta)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, gamma, arg, mean, rstd, scale, shift = ctx.saved_tensors
        R, K, C, training = ctx.cfg
        if not training:
            raise RuntimeError("pdfnet_amd: BatchNorm backward in eval mode is not implemented")
        g = dout.contiguous()
        dx = torch.empty_like(x)
        g_par, b_par = ctx.params
        mg_g, mg_b = _main_grad(g_par, g_par), _main_grad(b_par, b_par)
        direct = mg_g is not None and mg_b is not None
        dgamma = mg_g if direct else torch.empty(C, device=x.device)
        dbeta = mg_b if direct else torch.empty(C, device=x.device)
        L = _L()
        ws = _ws(_bn_ws_floats(C, R) + 3 * C, x.device)
        L.pdf_bn_relu_maxk_bwd(ptr(g), C, ptr(arg), ptr(x), C, ptr(mean), ptr(rstd), ptr(gamma), ptr(scale), ptr(shift), C, R, K,
                               ptr(dx), C, ptr(dgamma), ptr(dbeta), int(direct), ptr(ws), stream())
        if direct:
            dgamma = dbeta = None
        return dx, dgamma, dbeta, None, None, None, None, None, None


def bn_relu_max_over_k(x, gamma, beta, rmean, rvar, K, training, momentum=0.1, eps=1e-5):
    return _BnReluMaxK.apply(x, gamma, beta, rmean, rvar, K, training, momentum, eps)


# ----------------------------------------------------------------------------------------------
class _Cheby2(Function):
    """x [B,V,F] -> [B,V,2F] = interleave(x, L x); ell = (col, val, colT, valT, width)."""

    @staticmethod
    def forward(ctx, x, col, val, colT, valT):
        hip.require_gpu(x)
        x = x.contiguous()
        B, V, Fd = x.shape
        Wd = col.shape[1]
        out = torch.empty((B, V, 2 * Fd), device=x.device)
        _L().pdf_cheby2_fwd(ptr(x), Fd, B, V, Fd, ptr(col), ptr(val), Wd, ptr(out), 2 * Fd, stream())
        ctx.save_for_backward(colT, valT)
        ctx.cfg = (B, V, Fd)
        return out

    @staticmethod
    def backward(ctx, d):
        colT, valT = ctx.saved_tensors
        B, V, Fd = ctx.cfg
        dx = torch.empty((B, V, Fd), device=d.device)
        _L().pdf_cheby2_bwd(ptr(d.contiguous()), 2 * Fd, B, V, Fd, ptr(colT), ptr(valT), colT.shape[1], ptr(dx), Fd, stream())
        return dx, None, None, None, None


def cheby2(x, ell):
    return _Cheby2.apply(x, *ell)


class _Cheby2Pair(Function):
    """x [2,B,V,F] -> [2,B,V,2F]; half 0 with Laplacian ell0, half 1 with ell1 (same ELL width)."""

    @staticmethod
    def forward(ctx, x, col0, val0, colT0, valT0, col1, val1, colT1, valT1):
        hip.require_gpu(x)
        x = x.contiguous()
        _, B, V, Fd = x.shape
        if col0.shape != col1.shape or colT0.shape != colT1.shape:
            raise ValueError("pdfnet_amd: cheby2_pair wants both Laplacians padded to the same ELL width")
        out = torch.empty((2, B, V, 2 * Fd), device=x.device)
        _L().pdf_cheby2_fwd_pair(ptr(x), Fd, B, V, Fd, ptr(col0), ptr(val0), ptr(col1), ptr(val1), col0.shape[1], ptr(out), 2 * Fd, stream())
        ctx.save_for_backward(colT0, valT0, colT1, valT1)
        ctx.cfg = (B, V, Fd)
        return out

    @staticmethod
    def backward(ctx, d):
        colT0, valT0, colT1, valT1 = ctx.saved_tensors
        B, V, Fd = ctx.cfg
        dx = torch.empty((2, B, V, Fd), device=d.device)
        _L().pdf_cheby2_bwd_pair(ptr(d.contiguous()), 2 * Fd, B, V, Fd, ptr(colT0), ptr(valT0), ptr(colT1), ptr(valT1), colT0.shape[1],
                                 ptr(dx), Fd, stream())
        return (dx,) + (None,) * 8


def cheby2_pair(x, ell0, ell1):
    return _Cheby2Pair.apply(x, *ell0, *ell1)


class _Attention(Function):
    """softmax(q k^T / sqrt(dh)) v, heads = contiguous dh slices of the last axis; q,k,v [B,V,F]."""

    @staticmethod
    def forward(ctx, q, k, v, heads, pdrop, seed, kv_shift):
        hip.require_gpu(q, k, v)
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        B, V, Fd = q.shape
        dh = Fd // heads
        out = torch.empty_like(q)
        stat = torch.empty((B, heads, V, 2), device=q.device)
        _L().pdf_attn_fwd(ptr(q), ptr(k), ptr(v), Fd, B, V, heads, dh, kv_shift, pdrop, seed, ptr(step_counter(q.device)), ptr(out), Fd, ptr(stat), stream())
        ctx.save_for_backward(q, k, v, out, stat)
        ctx.cfg = (heads, pdrop, seed, kv_shift)
        return out

    @staticmethod
    def backward(ctx, do):
        q, k, v, out, stat = ctx.saved_tensors
        heads, pdrop, seed, kv_shift = ctx.cfg
        B, V, Fd = q.shape
        dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
        dvec = torch.empty((B, heads, V), device=q.device)
        _L().pdf_attn_bwd(ptr(q), ptr(k), ptr(v), Fd, ptr(out), ptr(do.contiguous()), Fd, ptr(stat), B, V, heads, Fd // heads,
                          kv_shift, pdrop, seed, ptr(step_counter(q.device)), ptr(dq), ptr(dk), ptr(dv), Fd, ptr(dvec), stream())
        return dq, dk, dv, None, None, None, None


def attention(q, k, v, heads, pdrop=0.0, training=False, kv_shift=0):
    """kv_shift: queries of sample b attend to keys / values of sample (b + kv_shift) % B (q, k, v flattened to [B, V, F])."""
    p = float(pdrop) if training else 0.0
    shp = q.shape
    if q.dim() > 3:
        q, k, v = (t.reshape(-1, shp[-2], shp[-1]) for t in (q, k, v))
    return _Attention.apply(q, k, v, heads, p, next_seed() if p > 0 else 0, int(kv_shift)).reshape(shp)


# ----------------------------------------------------------------------------------------------
def depth2pcl(depth, mask, K, valid, seed=None):
    """Depth map -> per-hand clouds on the GPU (reference depth2pcl, intaghand_encoder.py:369-491, batched).
    depth [B,1,R,R], mask [B,2,R,R] (right, left), K [B,3,3], valid [B,2] -> choose i64 [B,2,1024], cloud [B,2,1024,3], count."""
    hip.require_gpu(depth, mask)
    B, _, H, W = depth.shape
    d, m = depth.detach().contiguous(), mask.detach().contiguous()
    choose = torch.empty((B, 2, 1024), dtype=torch.int64, device=d.device)
    cloud = torch.empty((B, 2, 1024, 3), device=d.device)
    count = torch.empty((B, 2), dtype=torch.int32, device=d.device)
    _L().pdf_depth2pcl(ptr(d), ptr(m), ptr(K.detach().float().contiguous()), ptr(valid.detach().float().contiguous()), B, H, W,
                       next_seed() if seed is None else int(seed), ptr(choose), ptr(cloud), ptr(count), stream())
    return choose, cloud, count


def fps_reorder(cloud, choose, S1, S2, start1=None, start2=None):
    """The reference's `--sample_strategy FPS` (lib/opts.py:231; the block it keeps commented out at lib/datasets/interhand.py:857-900,
    in its `farthest_point_sampling_fast` variant): after the 1,024 points of a hand were drawn, they are REORDERED so that the first
    S1 are the level-1 farthest-point picks and, among those, the first S2 the level-2 picks -- PointNet++ takes "the first S points"
    as centroids (lib/utils/utils.py:143,156), so this turns random centroids into farthest-point centroids.  Per cloud:
        s1 = unique(fps(points, S1));  order = [s1 ascending, the other indices ascending];  points, choose = points[order], choose[order]
        s2 = unique(fps(points[:S1], S2));  the first S1 entries are reordered the same way.
    cloud [Bc,N,3] f32, choose [Bc,N] i64, start1 / start2 int32 [Bc] (the reference draws them at random; default 0) -> (cloud, choose).
    The picks come from pdf_fps (single-wave kernel for N <= 1,024); the index bookkeeping is a handful of torch ops (off the train path)."""
    hip.require_gpu(cloud, choose)
    Bc, N, _ = cloud.shape
    if not (0 < S2 <= S1 <= N):
        raise ValueError("fps_reorder: need 0 < S2 <= S1 <= N")
    dev = cloud.device

    def order_of(picks, n):
        m = torch.ones((Bc, n), dtype=torch.int64, device=dev)
        m.scatter_(1, picks.long(), 0)                               # 0 = picked (duplicates collapse like np.unique), 1 = other
        return torch.argsort(m * n + torch.arange(n, device=dev)[None], dim=1)    # [picked ascending, others ascending]
    o1 = order_of(fps(cloud, S1, start1), N)
    cloud = torch.gather(cloud, 1, o1[:, :, None].expand(-1, -1, cloud.shape[2]))
    choose = torch.gather(choose, 1, o1)
    o2 = order_of(fps(cloud[:, :S1].contiguous(), S2, start2), S1)
    o2 = torch.cat((o2, torch.arange(S1, N, device=dev)[None].expand(Bc, -1)), 1)
    cloud = torch.gather(cloud, 1, o2[:, :, None].expand(-1, -1, cloud.shape[2]))
    choose = torch.gather(choose, 1, o2)
    return cloud, choose


def nms_top1(hm):
    """hm [B,C,H,W] -> (ind int64 [B,C], score [B,C]): 5x5 NMS + top-1 per map (no gradient)."""
    hip.require_gpu(hm)
    h = hm.detach().contiguous()                    # plain NCHW copy of a tiny map
    B, C, H, W = h.shape
    ind = torch.empty((B, C), dtype=torch.int64, device=h.device)
    score = torch.empty((B, C), device=h.device)
    _L().pdf_nms_top1(ptr(h), B * C, H, W, ptr(ind), ptr(score), stream())
    return ind, score


class _ManoLBS(Function):
    """ManoLayer.forward (manolayer.py:257-334, use_pca=False) with its backward: (root_aa [B,3], pose_aa [B,45], shape [B,10],
    trans [B,3] | None) -> (verts [B,778,3], joints [B,21,3])."""

    @staticmethod
    def forward(ctx, root_aa, pose_aa, shape, trans, consts, left, center_idx):
        hip.require_gpu(root_aa)
        root_aa, pose_aa, shape = root_aa.contiguous(), pose_aa.contiguous(), shape.contiguous()
        trans = trans.contiguous() if trans is not None else None
        B = root_aa.shape[0]
        verts = torch.empty((B, 778, 3), device=root_aa.device)
        joints = torch.empty((B, 21, 3), device=root_aa.device)
        _L().pdf_mano_lbs_fwd(ptr(root_aa), ptr(pose_aa), ptr(shape), ptr(trans),
                              ptr(consts['v_template']), ptr(consts['shapedirs']), ptr(consts['posedirs']),
                              ptr(consts['J_regressor']), ptr(consts['weights']), B, int(left), center_idx, ptr(verts), ptr(joints), stream())
        ctx.save_for_backward(root_aa, pose_aa, shape)
        ctx.cfg = (consts, left, center_idx, trans is not None)
        ctx.set_materialize_grads(False)
        return verts, joints

    @staticmethod
    def backward(ctx, dverts, djoints):
        root_aa, pose_aa, shape = ctx.saved_tensors
        consts, left, center_idx, has_trans = ctx.cfg
        B = root_aa.shape[0]
        dev = root_aa.device
        droot, dpose = torch.empty((B, 3), device=dev), torch.empty((B, 45), device=dev)
        dshape = torch.empty((B, 10), device=dev) if ctx.needs_input_grad[2] else None
        dtrans = torch.empty((B, 3), device=dev) if has_trans and ctx.needs_input_grad[3] else None
        _L().pdf_mano_lbs_bwd(ptr(root_aa), ptr(pose_aa), ptr(shape), ptr(consts['v_template']), ptr(consts['shapedirs']), ptr(consts['posedirs']),
                              ptr(consts['J_regressor']), ptr(consts['weights']), B, int(left), center_idx,
                              ptr(dverts.contiguous() if dverts is not None else None), ptr(djoints.contiguous() if djoints is not None else None),
                              ptr(droot), ptr(dpose), ptr(dshape), ptr(dtrans), stream())
        return droot, dpose, dshape, dtrans, None, None, None


def mano_lbs(consts, root_aa, pose_aa, shape, trans=None, side='left', center_idx=None):
    """ManoLayer.forward (manolayer.py:257-334, use_pca=False), differentiable w.r.t. root / pose / shape / trans."""
    return _ManoLBS.apply(root_aa, pose_aa, shape, trans, consts, side == 'left', -1 if center_idx is None else int(center_idx))


class _ManoSplitCoeff(Function):
    """ManoRender.Split_coeff (Mano_render.py:145-194) for both hands: params map [B,122,H,W] (channels_last), ind [B,2], K [B,3,3]
    -> orient [2,B,3], pose [2,B,45], shape [2,B,10] (zeros), trans [2,B,3]."""

    @staticmethod
    def forward(ctx, params, ind, K, input_res, down):
        hip.require_gpu(params, ind)
        params = cl(params)
        B, C, H, W = params.shape
        if C != 122:
            raise ValueError("pdfnet_amd: the params head has 122 channels, got %d" % C)
        ind, K = ind.contiguous(), K.detach().float().contiguous()
        dev = params.device
        orient, pose = torch.empty((2, B, 3), device=dev), torch.empty((2, B, 45), device=dev)
        shape, trans = torch.empty((2, B, 10), device=dev), torch.empty((2, B, 3), device=dev)
        _L().pdf_mano_split_coeff(ptr(params), C, H * W, ptr(ind), ptr(K), B, input_res, down, ptr(orient), ptr(pose), ptr(shape), ptr(trans), stream())
        ctx.save_for_backward(params, ind, K)
        ctx.cfg = (input_res, down)
        ctx.mark_non_differentiable(shape)
        ctx.set_materialize_grads(False)
        return orient, pose, shape, trans

    @staticmethod
    def backward(ctx, do, dp, _ds, dt):
        params, ind, K = ctx.saved_tensors
        input_res, down = ctx.cfg
        B, C, H, W = params.shape
        z = lambda t, n: t.contiguous() if t is not None else torch.zeros((2, B, n), device=params.device)
        dparams = _zeros_cl(params.shape, params.device)
        _L().pdf_mano_split_coeff_bwd(ptr(params), ptr(dparams), C, H * W, ptr(ind), ptr(K), B, input_res, down,
                                      ptr(z(do, 3)), ptr(z(dp, 45)), ptr(z(dt, 3)), stream())
        return dparams, None, None, None, None


def mano_split_coeff(params, ind, K, input_res, down_ratio=4):
    return _ManoSplitCoeff.apply(params, ind, K, int(input_res), int(down_ratio))


class _RegressJoints(Function):
    """joints[b] = reg[J,778] @ verts[b]  (full_regressor, Mano_model.py:309-323; simplified.py:431-434)."""

    @staticmethod
    def forward(ctx, reg, verts):
        verts = verts.contiguous()
        B, Vn, _ = verts.shape
        J = reg.shape[0]
        out = torch.empty((B, J, 3), device=verts.device)
        _L().pdf_bmm_strided(ptr(reg), ptr(verts), ptr(out), B, J, 3, Vn, 1,
                             0, 0, Vn, 1, Vn * 3, 0, 3, 1, J * 3, 3, 1, 0, stream())
        ctx.save_for_backward(reg)
        return out

    @staticmethod
    def backward(ctx, g):
        (reg,) = ctx.saved_tensors
        g = g.contiguous()
        B, J, _ = g.shape
        Vn = reg.shape[1]
        dv = torch.empty((B, Vn, 3), device=g.device)
        # dv[b][v][c] = sum_j reg[j][v] g[b][j][c]
        _L().pdf_bmm_strided(ptr(reg), ptr(g), ptr(dv), B, Vn, 3, J, 1,
                             0, 0, 1, Vn, J * 3, 0, 3, 1, Vn * 3, 3, 1, 0, stream())
        return None, dv


def regress_joints(reg, verts):
    return _RegressJoints.apply(reg, verts)


class _RegressJointsPair(Function):
    """verts [2,B,V,3] -> joints [2,B,J,3]: half 0 with reg0, half 1 with reg1 (left / right full regressors)."""

    @staticmethod
    def forward(ctx, reg0, reg1, verts):
        verts = verts.contiguous()
        _, B, Vn, _ = verts.shape
        J = reg0.shape[0]
        out = torch.empty((2, B, J, 3), device=verts.device)
        for i, reg in enumerate((reg0, reg1)):
            _L().pdf_bmm_strided(ptr(reg), ptr(verts[i]), ptr(out[i]), B, J, 3, Vn, 1,
                                 0, 0, Vn, 1, Vn * 3, 0, 3, 1, J * 3, 3, 1, 0, stream())
        ctx.save_for_backward(reg0, reg1)
        return out

    @staticmethod
    def backward(ctx, g):
        regs = ctx.saved_tensors
        g = g.contiguous()
        _, B, J, _ = g.shape
        Vn = regs[0].shape[1]
        dv = torch.empty((2, B, Vn, 3), device=g.device)
        for i, reg in enumerate(regs):
            _L().pdf_bmm_strided(ptr(reg), ptr(g[i]), ptr(dv[i]), B, Vn, 3, J, 1,
                                 0, 0, 1, Vn, J * 3, 0, 3, 1, Vn * 3, 3, 1, 0, stream())
        return None, None, dv


def regress_joints_pair(reg0, reg1, verts):
    return _RegressJointsPair.apply(reg0, reg1, verts)


class _ProjectPoints(Function):
    """p[..., b, n, :] = pts[..., b, n, :] @ K[b]^T  (pts [G..., B, n, 3], K [B, 3, 3], no gradient to K): the pinhole matrix
    product of get_Landmarks_new (Mano_render.py:203-209) -- a HIP kernel instead of an aten (rocBLAS) batched matmul."""

    @staticmethod
    def forward(ctx, pts, K):
        hip.require_gpu(pts, K)
        pts, K = pts.contiguous(), K.detach().float().reshape(-1, 3, 3).contiguous()
        B, n = pts.shape[-3], pts.shape[-2]
        if pts.shape[-1] != 3 or K.shape[0] != B:
            raise ValueError("pdfnet_amd: project_points wants pts [..., B, n, 3] and K [B, 3, 3]")
        out = torch.empty_like(pts)
        G = pts.numel() // (B * n * 3)
        for h in range(G):
            _L().pdf_bmm_strided(ptr_at(pts, h * B * n * 3), ptr(K), ptr_at(out, h * B * n * 3), B, n, 3, 3, 1,
                                 n * 3, 0, 3, 1, 9, 0, 1, 3, n * 3, 3, 1, 0, stream())
        ctx.save_for_backward(K)
        ctx.cfg = (G, B, n)
        return out

    @staticmethod
    def backward(ctx, g):
        (K,) = ctx.saved_tensors
        G, B, n = ctx.cfg
        g = g.contiguous()
        d = torch.empty_like(g)
        for h in range(G):
            _L().pdf_bmm_strided(ptr_at(g, h * B * n * 3), ptr(K), ptr_at(d, h * B * n * 3), B, n, 3, 3, 1,
                                 n * 3, 0, 3, 1, 9, 0, 3, 1, n * 3, 3, 1, 0, stream())
        return d, None


def project_points(pts, K):
    return _ProjectPoints.apply(pts, K)


class _RowLoss(Function):
    """out[rows...] = mean over the trailing dims of |pred - tgt| (mode 0) or (pred - tgt)^2 (mode 1); gradient to pred only."""

    @staticmethod
    def forward(ctx, pred, tgt, row_dims, mode):
        hip.require_gpu(pred, tgt)
        pred, tgt = pred.contiguous(), tgt.contiguous()
        if pred.shape != tgt.shape:
            raise ValueError("pdfnet_amd: rowloss wants equal shapes, got %s vs %s" % (tuple(pred.shape), tuple(tgt.shape)))
        rows = 1
        for d in pred.shape[:row_dims]:
            rows *= d
        n = pred.numel() // max(rows, 1)
        out = torch.empty(pred.shape[:row_dims], dtype=torch.float32, device=pred.device)
        _L().pdf_rowloss_fwd(ptr(pred), ptr(tgt), rows, n, mode, ptr(out), stream())
        ctx.save_for_backward(pred, tgt)
        ctx.cfg = (rows, n, mode)
        return out

    @staticmethod
    def backward(ctx, g):
        pred, tgt = ctx.saved_tensors
        rows, n, mode = ctx.cfg
        dp = torch.empty_like(pred)
        _L().pdf_rowloss_bwd(ptr(pred), ptr(tgt), ptr(g.contiguous()), rows, n, mode, ptr(dp), stream())
        return dp, None, None, None


def rowloss(pred, tgt, row_dims, mode):
    """mode 'l1' | 'l2'."""
    return _RowLoss.apply(pred, tgt.detach(), row_dims, 0 if mode == 'l1' else 1)


class _FaceLoss(Function):
    """(normal_loss[G], edge_length_loss[G]) of lib/trains/simplified.py:66-115 for pred / gt [G,B,V,3] and faces [G,F,3] int64."""

    @staticmethod
    def forward(ctx, pred, gt, faces, edge_grad):
        hip.require_gpu(pred, gt, faces)
        pred, gt, faces = pred.contiguous(), gt.contiguous(), faces.contiguous()
        G, B, V, _ = pred.shape
        Fc = faces.shape[1]
        part = torch.empty((G, B, 2), dtype=torch.float32, device=pred.device)
        _L().pdf_face_loss_fwd(ptr(pred), ptr(gt), ptr(faces), G, B, V, Fc, ptr(part), stream())
        ctx.save_for_backward(pred, gt, faces)
        ctx.edge_grad = edge_grad
        ctx.set_materialize_grads(False)
        m = part.sum(1) / float(B * 3 * Fc)
        return m[:, 0], m[:, 1]

    @staticmethod
    def backward(ctx, gn, ge):
        pred, gt, faces = ctx.saved_tensors
        G, B, V, _ = pred.shape
        Fc = faces.shape[1]
        k = 1.0 / float(B * 3 * Fc)
        wn = (gn * k).contiguous() if gn is not None else torch.zeros(G, device=pred.device)
        we = (ge * k).contiguous() if (ge is not None and ctx.edge_grad) else None
        dp = torch.empty_like(pred)
        _L().pdf_face_loss_bwd(ptr(pred), ptr(gt), ptr(faces), G, B, V, Fc, ptr(wn), ptr(we), ptr(dp), stream())
        return dp, None, None, None


def face_loss(pred, gt, faces, edge_grad=True):
    return _FaceLoss.apply(pred, gt.detach(), faces, edge_grad)


class _DenseLoss(Function):
    """The three dense-map terms of CtdetLoss (simplified.py:368,374,376,391) in one forward and one backward launch:
    (SmoothL1(mask, mask_gt) scalar, MSE(hms, hms_gt) scalar, focal(clamp(sigmoid(hm)), hm_gt) [B])."""

    @staticmethod
    def forward(ctx, mask, mask_gt, hms, hms_gt, hm, hm_gt):
        hip.require_gpu(mask, hms, hm)
        mask, hms, hm = cl(mask), cl(hms), cl(hm)                       # the model's own layout: no copies on the product path
        mask_gt, hms_gt, hm_gt = (t.detach().float().contiguous() for t in (mask_gt, hms_gt, hm_gt))
        for a, b in ((mask, mask_gt), (hms, hms_gt), (hm, hm_gt)):
            if a.shape != b.shape:
                raise ValueError("pdfnet_amd: dense_loss wants equal shapes, got %s vs %s" % (tuple(a.shape), tuple(b.shape)))
        B = mask.shape[0]
        L = _L()
        ws = _ws(L.pdf_dense_loss_workspace_floats(B), mask.device)
        out = torch.empty(3 + 2 * B, dtype=torch.float32, device=mask.device)
        dims = lambda t: (t.shape[1], t.shape[2] * t.shape[3])
        L.pdf_dense_loss_fwd(ptr(mask), ptr(mask_gt), *dims(mask), ptr(hms), ptr(hms_gt), *dims(hms), ptr(hm), ptr(hm_gt), *dims(hm),
                             B, ptr(ws), ptr(out), stream())
        ctx.save_for_backward(mask, mask_gt, hms, hms_gt, hm, hm_gt, out)
        ctx.set_materialize_grads(False)
        return out[0], out[1], out[2:2 + B]

    @staticmethod
    def backward(ctx, g_mask, g_hms, g_hm):
        mask, mask_gt, hms, hms_gt, hm, hm_gt, out = ctx.saved_tensors
        B = mask.shape[0]
        g_mask, g_hms, g_hm = (t.contiguous() if t is not None else None for t in (g_mask, g_hms, g_hm))
        need = ctx.needs_input_grad
        dmask = torch.empty_like(mask) if need[0] and g_mask is not None else None
        dhms = torch.empty_like(hms) if need[2] and g_hms is not None else None
        dhm = torch.empty_like(hm) if need[4] and g_hm is not None else None
        dims = lambda t: (t.shape[1], t.shape[2] * t.shape[3])
        _L().pdf_dense_loss_bwd(ptr(mask), ptr(mask_gt), ptr(dmask), *dims(mask), ptr(hms), ptr(hms_gt), ptr(dhms), *dims(hms),
                                ptr(hm), ptr(hm_gt), ptr(dhm), *dims(hm), B, ptr(g_mask), ptr(g_hms), ptr(g_hm), ptr(out), stream())
        return dmask, None, dhms, None, dhm, None


def dense_loss(mask, mask_gt, hms, hms_gt, hm, hm_gt):
    return _DenseLoss.apply(mask, mask_gt, hms, hms_gt, hm, hm_gt)


def point_dist_sum(pred, gt):
    """pred, gt [..., n, dim] -> [...] sums over the n points of the Euclidean distance (evaluation metric, no gradient)."""
    hip.require_gpu(pred, gt)
    p, q = pred.detach().float().contiguous(), gt.detach().float().contiguous()
    if p.shape != q.shape:
        raise ValueError("pdfnet_amd: point_dist_sum wants equal shapes, got %s vs %s" % (tuple(p.shape), tuple(q.shape)))
    n, dim = p.shape[-2], p.shape[-1]
    out = torch.empty(p.shape[:-2], dtype=torch.float32, device=p.device)
    _L().pdf_point_dist_sum(ptr(p), ptr(q), out.numel(), n, dim, ptr(out), stream())
    return out


# ----------------------------------------------------------------------------------------------
# Fused mesh decoder (csrc/meshdec.hip, round 5): one DualGraphLayer (DualGraph.py:62-92) = three launches forward, see the kernel file.
MESH_FUSED = _os.environ.get("PDFNET_MESH_FUSED", "1") != "0"
# bf16 mode: the mesh decoder stays on the fused fp32 kernels (its products are latency-bound, not MFMA-bound: 1 % of the step's FLOPs; the
# fused form removes ~500 launches from a step that is bound by the host's issue time at B = 32).  0: the unfused bf16 GEMM chain of rounds 2-4.
MESH_FUSED_BF16 = _os.environ.get("PDFNET_MESH_FUSED_BF16", "1") != "0"
MESH_BF16_MFMA = _os.environ.get("PDFNET_MESH_BF16_MFMA", "1") != "0"       # bf16 mode: the fused levels' linears on the bf16 MFMA (0: fp32 math there)


def mesh_bf16_mfma():
    return _GEMM_BF16 and MESH_BF16_MFMA


def mesh_x3():
    """fp32 mode: the fused levels' linear products as x3 arithmetic (bit 2 of the library's x3 mode; PDF_X3_MESH=0 / set_x3 keep the native fp32 MFMA)."""
    return (not _GEMM_BF16) and (_L().pdf_debug_x3_mode() & 4) != 0


def _mesh_entry(L, which):
    if mesh_bf16_mfma():
        return getattr(L, 'pdf_mesh_level_%s_bf16' % which)
    if mesh_x3():
        return getattr(L, 'pdf_mesh_level_%s_x3' % which)
    return getattr(L, 'pdf_mesh_level_%s' % which)


def _pair(dst, l, r):
    dst[0], dst[1] = ptr(l), ptr(r)


def _mesh_lin(dst, l, r):
    _pair(dst.w, l.weight, r.weight)
    _pair(dst.b, l.bias, r.bias)


def mesh_level_params(layer):
    """The parameter tensors of a DualGraphLayer in the order of PdfMeshLevel: [(left, right) module pairs] per GCN block and attention block."""
    gl, gr = layer.graph_left.GCN_blocks, layer.graph_right.GCN_blocks
    at = layer.attn
    sl, sr = at.L_self_attn_layer, at.R_self_attn_layer
    gcn = [dict(fc1=(bl.fc1, br.fc1), fc2=(bl.fc2, br.fc2), sc=(bl.shortcut, br.shortcut), n2=(bl.norm2, br.norm2), n3=(bl.norm3, br.norm3))
           for bl, br in zip(gl, gr)]
    self_ = dict(ln=(sl.layer_norm, sr.layer_norm), q=(sl.w_qs, sr.w_qs), k=(sl.w_ks, sr.w_ks), v=(sl.w_vs, sr.w_vs), fc=(sl.fc, sr.fc),
                 ffln=(sl.ff.layer_norm, sr.ff.layer_norm), f1=(sl.ff.fc1, sr.ff.fc1), f2=(sl.ff.fc2, sr.ff.fc2))
    cross = dict(ln=(at.layer_norm1, at.layer_norm2), q=(at.w_qs, at.w_qs), k=(at.w_ks, at.w_ks), v=(at.w_vs, at.w_vs), fc=(at.fc, at.fc),
                 ffln=(at.ffL.layer_norm, at.ffR.layer_norm), f1=(at.ffL.fc1, at.ffR.fc1), f2=(at.ffL.fc2, at.ffR.fc2))
    return gcn, self_, cross


_MESH_GCN_KEYS = ('fc1', 'fc2', 'sc', 'n2', 'n3')
_MESH_ATT_KEYS = ('ln', 'q', 'k', 'v', 'fc', 'ffln', 'f1', 'f2')


def _mesh_args(layer, x, save, p, out, tape, qkv):
    """-> hip.MeshLevel filled for the forward of `layer` on x [2, B, V, cin].  save: write the tape the backward reads; p: dropout probability."""
    a = hip.MeshLevel()
    _, B, V, cin = x.shape
    level = {63: 0, 126: 1, 252: 2}.get(V)
    bl, br = layer.graph_left.GCN_blocks[0], layer.graph_right.GCN_blocks[0]
    if level is None or cin != 2 * (256 >> level) or bl.fc1.weight.shape[0] != (256 >> level) or len(layer.graph_left.GCN_blocks) != 4 or layer.attn.n_heads != 4:
        raise ValueError("pdfnet_amd: the fused mesh decoder covers the reference's levels (V = 63 / 126 / 252, C = 256 / 128 / 64, 4 blocks, 4 heads)")
    a.level, a.B, a.training, a.cin0, a.p = level, B, 1 if (save or p > 0) else 0, cin, p
    a.step = ptr(step_counter(x.device)) if p > 0 else None
    a.x, a.out, a.tape, a.qkv = ptr(x), ptr(out), ptr(tape), ptr(qkv)
    for k, (tl, tr) in zip(('ell_col', 'ell_val', 'ell_colT', 'ell_valT'), zip(bl.ell, br.ell)):
        _pair(getattr(a, k), tl, tr)
    a.ell_w = bl.ell_col.shape[1]
    gcn, self_, cross = mesh_level_params(layer)
    for i, blk in enumerate(gcn):
        for k in _MESH_GCN_KEYS:
            _mesh_lin(getattr(a.gcn[i], k), *blk[k])
        a.gcn[i].seed = next_seed() if p > 0 else 0          # (drawn in the order the unfused path draws them: same masks)
    for dst, src in ((a.self_, self_), (a.cross, cross)):
        for k in _MESH_ATT_KEYS:
            _mesh_lin(getattr(dst, k), *src[k])
        for k in ('seed_att', 'seed_z', 'seed_t', 'seed_x'):
            setattr(dst, k, next_seed() if p > 0 else 0)
    return a


def mesh_level_forward(layer, x, training=False, save=False):
    """One DualGraphLayer forward on the fused kernels, no autograd (x: [2, B, V, cin] with the position embedding added).
    -> (out [2, B, V, C], args, tape, qkv): the last three are what the backward needs."""
    hip.require_gpu(x)
    x = x.contiguous()
    _, B, V, cin = x.shape
    C = cin // 2
    level = {63: 0, 126: 1, 252: 2}[V]
    out = torch.empty((2, B, V, C), dtype=torch.float32, device=x.device)
    tape = torch.empty(_L().pdf_mesh_tape_floats(level, B), dtype=torch.float32, device=x.device)
    qkv = torch.empty((3, 2, B, V, C), dtype=torch.float32, device=x.device)
    a = _mesh_args(layer, x, save or training, float(layer.attn.p) if training else 0.0, out, tape, qkv)
    # bf16 mode: the build of the same kernels whose linear products run on the bf16 MFMA (csrc/meshdec_bf16.hip); same tape
    _mesh_entry(_L(), 'fwd')(_byref(a), stream())
    return out, a, tape, qkv


def _mesh_param_list(layer):
    """Every parameter tensor the level uses, once, in a fixed order; and for each (group, index, key, hand, 'w' | 'b') where its gradient goes."""
    gcn, self_, cross = mesh_level_params(layer)
    seen, tensors, slots = {}, [], []

    def add(t, where):
        if id(t) not in seen:
            seen[id(t)] = len(tensors)
            tensors.append(t)
            slots.append([])
        slots[seen[id(t)]].append(where)
    for i, blk in enumerate(gcn):
        for k in _MESH_GCN_KEYS:
            for hnd, m in enumerate(blk[k]):
                add(m.weight, ('ggcn', i, k, hnd, 'w'))
                add(m.bias, ('ggcn', i, k, hnd, 'b'))
    for grp, src in (('gself', self_), ('gcross', cross)):
        for k in _MESH_ATT_KEYS:
            for hnd, m in enumerate(src[k]):
                add(m.weight, (grp, None, k, hnd, 'w'))
                add(m.bias, (grp, None, k, hnd, 'b'))
    return tensors, slots


class _MeshLevel(Function):
    """One DualGraphLayer (DualGraph.py:62-92 after the position embedding) on the fused kernels: three launches forward, five backward plus
    the layer's 28 weight-gradient GEMMs on the weight-gradient side stream (csrc/meshdec.hip)."""

    @staticmethod
    def forward(ctx, x, layer, training, *params):
        need = any(ctx.needs_input_grad)                     # (grad mode is off inside Function.forward: ask the context)
        x = x.contiguous()
        out, a, tape, qkv = mesh_level_forward(layer, x, training, save=need)
        if need:
            ctx.layer, ctx.args, ctx.keep = layer, a, (x, tape, qkv, out)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, tape, qkv, out = ctx.keep
        a, layer = ctx.args, ctx.layer
        L = _L()
        dout = dout.contiguous()
        _, B, V, cin = x.shape
        C = cin // 2
        dx = torch.empty_like(x)
        gtape = torch.empty(L.pdf_mesh_gtape_floats(a.level, B), dtype=torch.float32, device=x.device)
        M = B * V
        nws = max(2 * L.pdf_wgrad_workspace_floats(M, C, 4 * C), 2 * L.pdf_wgrad_workspace_floats(M, C, 2 * C), 2 * L.pdf_wgrad_workspace_floats(M, C, C),
                  L.pdf_wgrad_workspace_floats(2 * M, C, C))
        ws = _ws(nws, x.device)
        a.dout, a.dx, a.gtape, a.wg_ws, a.wg_ws_floats = ptr(dout), ptr(dx), ptr(gtape), ptr(ws), nws
        tensors, slots = _mesh_param_list(layer)
        grads = []
        for t, where in zip(tensors, slots):
            g = _main_grad(t, t)
            ret = None
            if g is None:                                    # no flat gradient buffer behind this parameter: hand autograd a fresh one
                g = ret = torch.zeros_like(t)
            grads.append(ret)
            for grp, i, k, hnd, wb in where:
                dst = getattr(a, grp)
                if i is not None:
                    dst = dst[i]
                getattr(getattr(dst, k), wb)[hnd] = ptr(g)
        cur = stream()
        # The weight gradients go to the side stream only when they are accumulated straight into the trainer's flat buffer (nothing on the
        # chain reads them before join_wgrad).  Gradients handed back to autograd are consumed on THIS stream right away (AccumulateGrad, DDP's
        # bucket copy): they are produced on it too.
        direct = all(g is None for g in grads)
        with wgrad_stream(direct, x, tape, qkv, gtape, ws, dout, params=tensors):
            _mesh_entry(L, 'bwd')(_byref(a), cur, stream())
        ctx.keep = ctx.args = None
        return (dx, None, None) + tuple(grads)


def mesh_level(layer, x):
    """DualGraphLayer.forward after the position embedding, fused (x [2, B, V, cin])."""
    tensors, _ = _mesh_param_list(layer)
    return _MeshLevel.apply(x, layer, layer.training, *tensors)


def mesh_level_ok(layer, x):
    """The fused kernels cover the reference's three levels (V = 63 / 126 / 252 with C = 256 / 128 / 64, four blocks, four heads) in fp32 mode."""
    if not (MESH_FUSED and x.is_cuda and (MESH_FUSED_BF16 or not _GEMM_BF16) and x.dim() == 4 and x.shape[0] == 2 and x.dtype == torch.float32):
        return False
    level = {63: 0, 126: 1, 252: 2}.get(x.shape[2])
    return (level is not None and x.shape[3] == 2 * (256 >> level) and len(layer.graph_left.GCN_blocks) == 4 and layer.attn.n_heads == 4
            and layer.graph_left.GCN_blocks[0].fc1.weight.shape[0] == (256 >> level) and x.shape[1] <= 2048)


# ----------------------------------------------------------------------------------------------
# Fused mesh loss (csrc/loss.hip mesh_loss_*, round 5): every mesh term of CtdetLoss's train branch, two launches forward + one backward.
MESH_LOSS_FUSED = _os.environ.get("PDFNET_MESH_LOSS_FUSED", "1") != "0"
MESH_LOSS_TERMS = ('verts2d_loss', 'norm_loss', 'edge_loss', 'gcn_2d_loss', 'root_loss', 'verts_loss', 'abs_verts_loss', 'gcn_loss',
                   'abs_joints_loss', 'joints2d_loss', 'joints_loss', 'bone_direc_loss')


class _MeshLoss(Function):
    """(vp [2,B,778,3], v2p [2,B,778,2], hd3 [2,B,252,3], hd2 [2,B,252,2], r [2,B,3]) -> (weighted sum [B] of the 12 terms with `coefs`, the
    terms themselves [4 + 8 B] for the statistics, not differentiable)."""

    @staticmethod
    def forward(ctx, vp, v2p, hd3, hd2, r, gt, consts, size, down, edge_grad, coefs):
        hip.require_gpu(vp)
        vp, v2p, hd3, hd2, r = (t.contiguous() for t in (vp, v2p, hd3, hd2, r))
        B = vp.shape[1]
        if vp.shape[2:] != (778, 3) or hd3.shape[2:] != (252, 3):
            raise ValueError("pdfnet_amd: mesh_loss wants the decoder's 778- and 252-vertex meshes")
        a = hip.MeshLoss()
        keep = [vp, v2p, hd3, hd2, r]
        for k, t in (('vp', vp), ('v2p', v2p), ('hd3', hd3), ('hd2', hd2), ('r', r)):
            setattr(a, k, ptr(t))
        for k in ('vgt', 'jgt', 'v2gt', 'lmsgt'):               # (left, right) pairs of the batch's own tensors: nothing is stacked
            for i in (0, 1):
                t = gt[k][i].contiguous()
                keep.append(t)
                getattr(a, k)[i] = ptr(t)
        for k in ('ind', 'K', 'valid'):
            t = gt[k].contiguous()
            keep.append(t)
            setattr(a, k, ptr(t))
        regs, faces, perms = consts
        a.reg[0], a.reg[1] = ptr(regs[0]), ptr(regs[1])
        a.faces = ptr(faces)
        a.perm[0], a.perm[1] = ptr(perms[0]), ptr(perms[1])
        a.B, a.Fc, a.size, a.down, a.edge_grad = B, faces.shape[1], int(size), int(down), 1 if edge_grad else 0
        for i, c in enumerate(coefs):
            a.coef[i] = float(c)
        part = torch.empty((2, B, 12), dtype=torch.float32, device=vp.device)
        out = torch.empty(4 + 9 * B, dtype=torch.float32, device=vp.device)
        a.part, a.out = ptr(part), ptr(out)
        _L().pdf_mesh_loss_fwd(_byref(a), stream())
        ctx.args, ctx.keep = a, keep + [regs, faces, perms, part]
        terms = out[:4 + 8 * B]
        ctx.mark_non_differentiable(terms)
        return out[4 + 8 * B:], terms

    @staticmethod
    def backward(ctx, g, _):
        a = ctx.args
        vp, v2p, hd3, hd2, r = ctx.keep[:5]
        g = g.contiguous()
        dvp, dv2p, dhd3, dhd2, dr = (torch.empty_like(t) for t in (vp, v2p, hd3, hd2, r))
        a.gmp = ptr(g)
        a.dvp, a.dv2p, a.dhd3, a.dhd2, a.dr = ptr(dvp), ptr(dv2p), ptr(dhd3), ptr(dhd2), ptr(dr)
        _L().pdf_mesh_loss_bwd(_byref(a), stream())
        return dvp, dv2p, dhd3, dhd2, dr, None, None, None, None, None, None


def mesh_loss(vp, v2p, hd3, hd2, r, gt, consts, size, down, edge_grad, coefs):
    """gt: dict vgt / jgt / v2gt / lmsgt = (left, right) pairs of [B,778,3] / [B,21,3] / [B,778,2] / [B,21,2], ind [B,2] int64, K [B,3,3], valid [B,2];
    consts: ((reg_left, reg_right) [21,778], faces [2,F,3] int64, (perm_left, perm_right) [1008] int64); coefs: the weights of the twelve
    terms in MESH_LOSS_TERMS order.  -> (their weighted sum per sample [B] (differentiable), dict of the twelve terms (detached))."""
    mp, terms = _MeshLoss.apply(vp, v2p, hd3, hd2, r, gt, consts, size, down, edge_grad, tuple(coefs))
    B = vp.shape[1]
    d = {k: terms[i] for i, k in enumerate(MESH_LOSS_TERMS[:4])}
    d.update({k: terms[4 + i * B:4 + (i + 1) * B] for i, k in enumerate(MESH_LOSS_TERMS[4:])})
    return mp, d
