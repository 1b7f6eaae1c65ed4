"""ctypes binding of libpdfnet_hip.so -- the drop-in C-ABI boundary (include/pdfnet_hip.h).

The product path has NO fallback: if the library is missing, import fails loudly; if a call returns
non-zero, a RuntimeError is raised (the reference's only native op calls exit(-1) on a launch error,
lib/utils/roi_align/src/cuda/crop_and_resize_kernel.cu:186-191 -- we raise instead).
torch is used only for device memory (tensor.data_ptr()) and the current HIP stream handle.
"""
import ctypes
import os
import re

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PDFNET_HIP_LIB") or os.path.join(_HERE, "libpdfnet_hip.so")   # override: A/B of two builds in one session
HEADER_PATH = os.path.join(_HERE, "..", "include", "pdfnet_hip.h")

_CT = {"int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float,
       "unsigned long long": ctypes.c_ulonglong}


def parse_header(path=HEADER_PATH):
    """-> {name: (restype, [argtypes])} for every `int|long pdf_*(...)` prototype in the header."""
    txt = open(path).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(int|long)\s+(pdf_\w+)\s*\(([^)]*)\)\s*;", txt):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        types = []
        for a in args.split(","):
            a = " ".join(a.split())
            if a in ("", "void"):
                continue
            if "*" in a:
                types.append(ctypes.c_void_p)
            else:
                base = a.rsplit(" ", 1)[0]
                types.append(_CT[base])
        protos[name] = (_CT[ret], types)
    return protos


class CallOpts(ctypes.Structure):
    """PdfCallOpts of include/pdfnet_hip.h: the explicit options of the `_x` entry points (field for field)."""
    _fields_ = [("op0_bf16", ctypes.c_void_p), ("op1_bf16", ctypes.c_void_p), ("out_bf16", ctypes.c_void_p), ("bn_x_bf16", ctypes.c_void_p),
                ("stats_out", ctypes.c_void_p), ("stats_cap", ctypes.c_long), ("stats_tiles", ctypes.c_long), ("stats_rows", ctypes.c_long),
                ("tile_stats", ctypes.c_void_p), ("tile_n", ctypes.c_long), ("tile_rows", ctypes.c_long),
                ("in_scale", ctypes.c_void_p), ("in_shift", ctypes.c_void_p), ("op1_bf16_t", ctypes.c_void_p), ("ws", ctypes.c_void_p), ("ws_floats", ctypes.c_long),
                ("wino_v", ctypes.c_void_p)]


_P2 = ctypes.c_void_p * 2


class MeshLin(ctypes.Structure):
    """PdfMeshLin / PdfMeshLinG / PdfMeshLN / PdfMeshLNG of include/pdfnet_hip.h: a (weight, bias) or (gamma, beta) pair per hand."""
    _fields_ = [("w", _P2), ("b", _P2)]


class MeshGcn(ctypes.Structure):
    _fields_ = [("fc1", MeshLin), ("fc2", MeshLin), ("sc", MeshLin), ("n2", MeshLin), ("n3", MeshLin), ("seed", ctypes.c_ulonglong)]


class MeshAttn(ctypes.Structure):
    _fields_ = [("ln", MeshLin), ("q", MeshLin), ("k", MeshLin), ("v", MeshLin), ("fc", MeshLin), ("ffln", MeshLin), ("f1", MeshLin), ("f2", MeshLin),
                ("seed_att", ctypes.c_ulonglong), ("seed_z", ctypes.c_ulonglong), ("seed_t", ctypes.c_ulonglong), ("seed_x", ctypes.c_ulonglong)]


class MeshGcnG(ctypes.Structure):
    _fields_ = [("fc1", MeshLin), ("fc2", MeshLin), ("sc", MeshLin), ("n2", MeshLin), ("n3", MeshLin)]


class MeshAttnG(ctypes.Structure):
    _fields_ = [("ln", MeshLin), ("q", MeshLin), ("k", MeshLin), ("v", MeshLin), ("fc", MeshLin), ("ffln", MeshLin), ("f1", MeshLin), ("f2", MeshLin)]


class MeshLevel(ctypes.Structure):
    """PdfMeshLevel of include/pdfnet_hip.h (field for field; the library's sizeof is checked at load)."""
    _fields_ = [("level", ctypes.c_int), ("B", ctypes.c_int), ("training", ctypes.c_int), ("cin0", ctypes.c_int), ("p", ctypes.c_float),
                ("step", ctypes.c_void_p), ("x", ctypes.c_void_p), ("out", ctypes.c_void_p),
                ("ell_col", _P2), ("ell_val", _P2), ("ell_colT", _P2), ("ell_valT", _P2), ("ell_w", ctypes.c_int),
                ("gcn", MeshGcn * 4), ("self_", MeshAttn), ("cross", MeshAttn),
                ("tape", ctypes.c_void_p), ("qkv", ctypes.c_void_p), ("dout", ctypes.c_void_p), ("dx", ctypes.c_void_p), ("gtape", ctypes.c_void_p),
                ("ggcn", MeshGcnG * 4), ("gself", MeshAttnG), ("gcross", MeshAttnG), ("wg_ws", ctypes.c_void_p), ("wg_ws_floats", ctypes.c_long)]


class MeshLoss(ctypes.Structure):
    """PdfMeshLoss of include/pdfnet_hip.h (field for field)."""
    _fields_ = [(k, ctypes.c_void_p) for k in ("vp", "v2p", "hd3", "hd2", "r")] + [(k, _P2) for k in ("vgt", "jgt", "v2gt", "lmsgt")] + \
               [(k, ctypes.c_void_p) for k in ("ind", "K", "valid")] + \
               [("reg", _P2), ("faces", ctypes.c_void_p), ("perm", _P2), ("B", ctypes.c_int), ("Fc", ctypes.c_int), ("size", ctypes.c_int),
                ("down", ctypes.c_int), ("part", ctypes.c_void_p), ("out", ctypes.c_void_p), ("coef", ctypes.c_float * 12), ("gmp", ctypes.c_void_p),
                ("edge_grad", ctypes.c_int)] + \
               [(k, ctypes.c_void_p) for k in ("dvp", "dv2p", "dhd3", "dhd2", "dr")]


class _Lib:
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "pdfnet_amd: %s is missing -- build it with `python -m pdfnet_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU or PyTorch fallback." % LIB_PATH)
        self.cdll = ctypes.CDLL(LIB_PATH)
        self.protos = parse_header()
        for name, (ret, args) in self.protos.items():
            fn = getattr(self.cdll, name)          # AttributeError if the .so does not export it
            fn.restype = ret
            fn.argtypes = args
        if self.cdll.pdf_debug_callopts_size() != ctypes.sizeof(CallOpts):
            raise ImportError("pdfnet_amd: PdfCallOpts of %s has %d bytes, this binding's has %d -- rebuild the library"
                              % (LIB_PATH, self.cdll.pdf_debug_callopts_size(), ctypes.sizeof(CallOpts)))
        if self.cdll.pdf_debug_mesh_loss_size() != ctypes.sizeof(MeshLoss):
            raise ImportError("pdfnet_amd: PdfMeshLoss of %s has %d bytes, this binding's has %d -- rebuild the library"
                              % (LIB_PATH, self.cdll.pdf_debug_mesh_loss_size(), ctypes.sizeof(MeshLoss)))
        if self.cdll.pdf_debug_mesh_level_size() != ctypes.sizeof(MeshLevel):
            raise ImportError("pdfnet_amd: PdfMeshLevel of %s has %d bytes, this binding's has %d -- rebuild the library"
                              % (LIB_PATH, self.cdll.pdf_debug_mesh_level_size(), ctypes.sizeof(MeshLevel)))

    def __getattr__(self, name):
        fn = getattr(self.cdll, name)
        is_status = self.protos[name][0] is ctypes.c_int and not name.startswith("pdf_debug_")     # int = status code

        def call(*args):
            rc = fn(*args)
            if is_status and rc != 0:
                raise RuntimeError("libpdfnet_hip: %s failed with code %d" % (name, rc))
            if _tape is not None and is_status:                # (taped.TapedSegment is recording: launches only, not size queries)
                _tape.append((call, args))
            return rc
        setattr(self, name, call)
        return call


_lib = None
_tape = None            # a list while pdfnet_amd.taped records a segment: every status-returning library call is appended as (callable, args)


def lib():
    global _lib
    if _lib is None:
        _lib = _Lib()
        if torch.cuda.is_available():
            # The library's rings live on the device that is current NOW (one process = one GPU).  Under a multi-GPU
            # launcher the rank must have selected its GPU first: binding the library before
            # torch.cuda.set_device(LOCAL_RANK) would put every rank's rings on GPU 0.
            lr = os.environ.get("LOCAL_RANK")
            if lr is not None and torch.cuda.device_count() > 1 and int(lr) != torch.cuda.current_device():
                _lib = None
                raise RuntimeError("pdfnet_amd: LOCAL_RANK=%s but the current device is cuda:%d -- call torch.cuda.set_device(LOCAL_RANK) "
                                   "(trains.base_trainer.init_distributed) before the first pdfnet_amd call" % (lr, torch.cuda.current_device()))
            _lib.pdf_init()                # device-side ticket counters: allocate now, never inside a stream capture
    return _lib


def check_device(dev):
    """Raise if tensors on `dev` would be handed to a library whose rings live on another GPU."""
    L = lib()
    have = L.pdf_debug_init_device()
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if have >= 0 and have != idx:
        raise RuntimeError("pdfnet_amd: libpdfnet_hip was initialised on cuda:%d, the model lives on cuda:%d (one process per GPU)" % (have, idx))


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def stream():
    """hipStream_t of torch's current stream (launches are captured when that stream is capturing).
    Called once per kernel launch (~2,000 times a step): the raw-handle query costs ~0.3 us where
    `torch.cuda.current_stream()` builds a Stream object through the device-index helpers (~9 us, a fifth of the step's
    host time)."""
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    """Device address as a plain int (ctypes converts ints and None for `void*` parameters itself; ~7,000 pointers a step)."""
    return t.data_ptr() if t is not None else None


def ptr_at(t, elements):
    """Device pointer `elements` elements into `t` (a channel offset inside an NHWC row)."""
    return t.data_ptr() + elements * t.element_size()


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("pdfnet_amd ops run only on the GPU (HIP); got a %s tensor. "
                               "There is no CPU fallback in the product path." % t.device)
